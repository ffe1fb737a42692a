#!/bin/bash
# r06_kin.sh -- the step kernel's stage clock with two more stamps (terms formed, every K word in) + the bitwise A/B of the kernel
O=gpurun_out/r06kin; mkdir -p $O
L="-I include -L hydrochrono_amd/lib -Wl,-rpath,$PWD/hydrochrono_amd/lib"
timeout 900 python -m pytest tests/test_gpu_boundary.py -x -q -m gpu -k "step_kernel_of_the_common_block_step" 2>&1 | tail -3
g++ -O2 -std=c++17 profiles/step_stamps_probe.cpp $L -lhydrochrono_amd_tuning -o /tmp/stamps && {
  echo "=== step_hot_kernel, gap 0"; /tmp/stamps 0 0
  echo "=== step_hot_kernel, gap 0, second run"; /tmp/stamps 0 0
  echo "=== step_hot_kernel, HC_STEP_HALVES=2, gap 0"; HC_STEP_HALVES=2 /tmp/stamps 0 0
  echo "=== step_hot_kernel, 100 us gaps"; /tmp/stamps 100 0
  echo "=== step_hot_kernel, ring in device memory"; HC_QUEUE_DEV_MEM=1 /tmp/stamps 0 0
} > $O/step_stage_clock_k_in.txt 2>&1
grep -E "===|as the caller|first entry|wave 0" $O/step_stage_clock_k_in.txt
awk '/=== step_hot_kernel, gap 0$/{f=1} f&&/stage  /{p=1} p{print} /dispatches/{if(p){exit}}' $O/step_stage_clock_k_in.txt
g++ -O2 -std=c++17 profiles/host_path_c.cpp $L -lhydrochrono_amd -o /tmp/host_path_c && { /tmp/host_path_c; /tmp/host_path_c; } > $O/host_path_c.txt 2>&1; cat $O/host_path_c.txt
