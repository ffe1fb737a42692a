#!/bin/bash
# r06_soak3.sh -- the final build once more: two suites back to back, then the SHIPPED library under the three draws of the differential run
O=gpurun_out/r06soak3; mkdir -p $O
for i in 1 2; do timeout 1300 python -m pytest tests -m gpu -q > $O/pytest_gpu_$i.txt 2>&1; tail -2 $O/pytest_gpu_$i.txt | cut -c1-200; done
FUZZ_RELEASE=1 timeout 400 python profiles/fuzz_parity.py 240 750001 > $O/fuzz_release.txt 2>&1; tail -1 $O/fuzz_release.txt | cut -c1-200
FUZZ_RELEASE=1 FUZZ_SHARDS=1 timeout 400 python profiles/fuzz_parity.py 200 760001 > $O/fuzz_release_shards.txt 2>&1; tail -1 $O/fuzz_release_shards.txt | cut -c1-200
FUZZ_RELEASE=1 FUZZ_WIDE=1 timeout 300 python profiles/fuzz_parity.py 120 770001 > $O/fuzz_release_wide.txt 2>&1; tail -1 $O/fuzz_release_wide.txt | cut -c1-200
