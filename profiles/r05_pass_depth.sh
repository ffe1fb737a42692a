#!/bin/bash
# round 5: the pass kernel alone at depth 16 / 32 / 64 over the K of C3 and of one C4/8 rank (tuning build, hc_tuning_time_pass):
# launches back to back (sustained load) and with 500 / 3000 us of idle GPU in front of each launch (what a block's worth of steps
# leaves between two passes in the product), then under rocprofv3 --kernel-trace --stats for the kernel rows.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
for pause in 0 500 3000; do
  echo "== HC_TUNING_PASS_PAUSE_US=$pause"
  HC_TUNING_PASS_PAUSE_US=$pause python profiles/pass_depth_probe.py 2>/dev/null
done > $O/pass_depth_probe.txt
cat $O/pass_depth_probe.txt
cd /tmp && export TMPDIR=/tmp
export HC_TUNING_PASS_PAUSE_US=500
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_pass_depth -- python3 $R/profiles/pass_depth_probe.py > $O/stats_pass_depth.log 2>&1
cp $(ls -t $O/stats_pass_depth/*/*kernel_stats.csv | head -1) $O/pass_depth_probe_kernel_stats.csv
grep conv_block $O/pass_depth_probe_kernel_stats.csv
