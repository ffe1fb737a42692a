#!/bin/bash
# round 5: the slices of a pass one block ahead WITHOUT the barrier bit between them (a slice's workgroups start while its predecessor's
# stragglers still run) against ordered slices (HC_SLICE_OVERLAP=0, tuning build): a C4/8 rank and C3 by caller gap, C++ caller.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
g++ -O2 -std=c++17 profiles/ahead_probe.cpp -I include -L hydrochrono_amd/lib -l:libhydrochrono_amd_tuning.so -Wl,-rpath,$R/hydrochrono_amd/lib -o /tmp/ahead_probe_t || exit 1
for ov in 0 1 0 1; do
  echo "== HC_SLICE_OVERLAP=$ov, rows of 64 of 512 bodies"
  HC_SLICE_OVERLAP=$ov SHARD_ROWS=64 /tmp/ahead_probe_t 1 2>/dev/null | grep "schedule  1" | cut -c1-140
done > $O/slice_overlap.txt
for ov in 0 1; do
  echo "== HC_SLICE_OVERLAP=$ov, C3"
  HC_SLICE_OVERLAP=$ov /tmp/ahead_probe_t 0 2>/dev/null | grep "schedule  1" | cut -c1-140
done >> $O/slice_overlap.txt
cat $O/slice_overlap.txt
