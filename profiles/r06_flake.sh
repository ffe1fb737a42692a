#!/bin/bash
# r06_flake.sh -- does test_a_failed_step_of_a_wide_context... time out again?  (it did once, in a full-suite run)
O=gpurun_out/r06flake; mkdir -p $O
for i in $(seq 1 14); do
  HC_STEP_TIMEOUT_S=5 timeout 300 python -m pytest tests/test_gpu_boundary.py -x -q -m gpu -k "failed_step_of_a_wide_context" 2>&1 | tail -1 >> $O/loop.txt
done
cat $O/loop.txt
timeout 900 python -m pytest tests/test_gpu_boundary.py tests/test_chrono_adapter.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
