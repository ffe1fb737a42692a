#!/bin/bash
# round 5: the depth-64 pass with two workgroups per CU (conv_block_kernel<3, 2, 4, 2>: 212 registers, two waves per SIMD) beside <3, 3, 4, 1>
# and <6, 3, 4, 1>: parity, then the kernel alone at C3 and one C4/8 rank.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
echo "== parity: HC_BLOCK64_MT=3 HC_BLOCK64_R=2"
HC_BLOCK64_MT=3 HC_BLOCK64_R=2 timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "depth_64 or depth64" 2>&1 | tail -3
for cfg in "3 2" "3 3" "6 3"; do
  set -- $cfg
  for pause in 500 0; do
    echo "== HC_BLOCK64_MT=$1 HC_BLOCK64_R=$2 HC_TUNING_PASS_PAUSE_US=$pause"
    HC_BLOCK64_MT=$1 HC_BLOCK64_R=$2 HC_TUNING_PASS_PAUSE_US=$pause python profiles/pass_depth_probe.py 2>/dev/null | grep "depth 64"
  done
done
