set -x
R=$GRAFT_REPO_ROOT
python $R/profiles/host_path.py > $R/gpurun_out/host_path.json 2> $R/gpurun_out/host_path.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof2_stats -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $R/gpurun_out/prof2_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof2_fetch -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof2_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof2_write -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof2_write.log 2>&1
