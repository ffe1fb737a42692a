#!/bin/bash
# round 5: one C4/8 rank through the Python wrapper, back to back, under the two schedules and the adaptive default, with and without waves
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
NT=512 python profiles/c4_share_sched.py 2>/dev/null > $O/c4_share_sched.txt
cat $O/c4_share_sched.txt
