#!/bin/bash
# r06_halves.sh -- step_hot_kernel: (a) requests for term slots past n_terms ask for the last live slot again (22 KB less per workgroup
# and step at C3), (b) HC_STEP_HALVES=2: two workgroups per row tile.  Stage clock + caller's median of each, and the bitwise A/B.
O=gpurun_out/r06halves5; mkdir -p $O
L="-I include -L hydrochrono_amd/lib -Wl,-rpath,$PWD/hydrochrono_amd/lib"
timeout 900 python -m pytest tests/test_gpu_boundary.py -x -q -m gpu -k "step_kernel_of_the_common_block_step" > $O/pytest_hot.txt 2>&1; tail -3 $O/pytest_hot.txt
g++ -O2 -std=c++17 profiles/step_stamps_probe.cpp $L -lhydrochrono_amd_tuning -o /tmp/stamps && {
  for i in 1 2; do
    echo "=== step_hot_kernel, one workgroup per row tile (24 + 1), gap 0, run $i"; /tmp/stamps 0 0
    echo "=== step_hot_kernel, HC_STEP_HALVES=2 (48 + 1 workgroups), gap 0, run $i"; HC_STEP_HALVES=2 /tmp/stamps 0 0
  done
  echo "=== one workgroup per row tile, 100 us gaps"; /tmp/stamps 100 0
  echo "=== HC_STEP_HALVES=2, 100 us gaps"; HC_STEP_HALVES=2 /tmp/stamps 100 0
  echo "=== one workgroup per row tile, ring in device memory"; HC_QUEUE_DEV_MEM=1 /tmp/stamps 0 0
  echo "=== HC_STEP_HALVES=2, ring in device memory"; HC_QUEUE_DEV_MEM=1 HC_STEP_HALVES=2 /tmp/stamps 0 0
} > $O/step_stage_clock_halves.txt 2>&1
grep -E "===|as the caller|doorbell ->|first entry|first -> last" $O/step_stage_clock_halves.txt
g++ -O2 -std=c++17 profiles/host_path_c.cpp $L -lhydrochrono_amd_tuning -o /tmp/host_path_c_t && {
  for i in 1 2; do echo "== halves 1"; /tmp/host_path_c_t; echo "== halves 2"; HC_STEP_HALVES=2 /tmp/host_path_c_t; done; } > $O/host_path_c_halves.txt 2>&1
cat $O/host_path_c_halves.txt
