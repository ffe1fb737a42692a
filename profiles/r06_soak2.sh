#!/bin/bash
# r06_soak2.sh -- after the step kernel's rewrite: the suite, then long draws on the SHIPPED library (every glitch of the day was there)
O=gpurun_out/r06soak2; mkdir -p $O
timeout 1300 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -6 $O/pytest_gpu.txt | cut -c1-300
FUZZ_RELEASE=1 timeout 500 python profiles/fuzz_parity.py 300 730001 > $O/fuzz_release.txt 2>&1; tail -2 $O/fuzz_release.txt | cut -c1-300
FUZZ_RELEASE=1 FUZZ_SHARDS=1 timeout 400 python profiles/fuzz_parity.py 200 740001 > $O/fuzz_release_shards.txt 2>&1; tail -2 $O/fuzz_release_shards.txt | cut -c1-300
