#!/bin/bash
# Round 6: the randomised differential run and the soak of the adaptive schedule on the round's library (step_hot_kernel, packet rings in
# device memory, enqueue_step split), both builds.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06soak; rm -rf $O; mkdir -p $O
timeout 400 python profiles/fuzz_parity.py 300 620001 > $O/fuzz_parity.txt 2>&1; tail -2 $O/fuzz_parity.txt
FUZZ_RELEASE=1 timeout 400 python profiles/fuzz_parity.py 300 630001 > $O/fuzz_parity_release.txt 2>&1; tail -2 $O/fuzz_parity_release.txt
FUZZ_WIDE=1 timeout 300 python profiles/fuzz_parity.py 180 640001 > $O/fuzz_parity_wide.txt 2>&1; tail -2 $O/fuzz_parity_wide.txt
FUZZ_SHARDS=1 timeout 300 python profiles/fuzz_parity.py 180 650001 > $O/fuzz_parity_shards.txt 2>&1; tail -2 $O/fuzz_parity_shards.txt
FUZZ_RELEASE=1 FUZZ_WIDE=1 timeout 300 python profiles/fuzz_parity.py 150 660001 > $O/fuzz_parity_release_wide.txt 2>&1; tail -2 $O/fuzz_parity_release_wide.txt
timeout 900 python profiles/soak_adaptive.py > $O/soak_adaptive_12000_steps.txt 2>&1; tail -3 $O/soak_adaptive_12000_steps.txt
timeout 900 python -m pytest tests/test_gpu_ahead.py tests/test_gpu_boundary.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/pytest_subset.txt 2>&1; tail -3 $O/pytest_subset.txt
