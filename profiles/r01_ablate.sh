R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for m in waves hs nowave_steps; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abl_$m -- python3 $R/profiles/step_ablation.py $m > /dev/null 2>&1
  echo "== $m"; grep -E "conv_step|finalize|conv_block" $R/gpurun_out/abl_$m/*/*_kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
done
