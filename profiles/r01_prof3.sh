R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof3_stats -- python3 $R/bench.py --steps 160 --warmup 16 --no-cpu-baseline --profile-stride 1000000 > $R/gpurun_out/prof3_stats.log 2>&1
tail -1 $R/gpurun_out/prof3_stats.log | cut -c1-160
