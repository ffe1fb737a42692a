#!/bin/bash
# round 5: second bisection of the depth-64 pass (interleaved form, HC_BLOCK64_R=11) -- 16: the gathers stay on 8 columns (cache hits),
# 17: K loads without the non-temporal hint, 18: the K loads of a fragment from 6 KB in a row (one stream per wave instead of six).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
for r in 11 16 17 18; do
  echo "== HC_BLOCK64_MT=6 HC_BLOCK64_R=$r HC_TUNING_PASS_PAUSE_US=500"
  HC_BLOCK64_MT=6 HC_BLOCK64_R=$r HC_TUNING_PASS_PAUSE_US=500 python profiles/pass_depth_probe.py 2>/dev/null | grep "depth 64"
done > $O/depth64_bisect2.txt 2>&1
cat $O/depth64_bisect2.txt
