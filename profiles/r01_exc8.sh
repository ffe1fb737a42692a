R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export HC_EXC_CHUNK_GP=8
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_exc8 -- python3 $R/bench.py --steps 64 --warmup 16 --no-cpu-baseline --profile-stride 1000000 > $R/gpurun_out/prof_exc8.log 2>&1
head -6 $R/gpurun_out/prof_exc8/*/*_kernel_stats.csv
