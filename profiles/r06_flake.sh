#!/bin/bash
# r06_flake.sh -- how often does test_a_failed_step_of_a_wide_context... time out?  (once, in one full-suite run of round 6)
O=gpurun_out/r06flake; mkdir -p $O; : > $O/loop.txt
for i in $(seq 1 ${LOOPS:-90}); do
  HC_STEP_TIMEOUT_S=3 timeout 300 python -m pytest tests/test_gpu_boundary.py -x -q -m gpu -k "failed_step_of_a_wide_context" > $O/last.txt 2>&1
  tail -1 $O/last.txt >> $O/loop.txt
  if ! tail -1 $O/last.txt | grep -q "1 passed"; then cp $O/last.txt $O/fail_$i.txt; fi
done
sort $O/loop.txt | cut -c1-20 | uniq -c
ls $O
