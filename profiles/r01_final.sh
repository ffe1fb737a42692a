# Final round-1 measurements: bench lines, rocprofv3 kernel stats, PMC traffic (separate passes).
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/final
mkdir -p $R/gpurun_out/final
python $R/bench.py > $R/gpurun_out/final/bench_c3.log 2>&1; tail -1 $R/gpurun_out/final/bench_c3.log > $R/gpurun_out/final/bench_c3.json
python $R/bench.py --lookahead 0 --no-cpu-baseline > $R/gpurun_out/final/bench_c3_plain.log 2>&1; tail -1 $R/gpurun_out/final/bench_c3_plain.log > $R/gpurun_out/final/bench_c3_plain.json
python $R/bench.py --scaling strong --bodies 512 --steps 64 --warmup 16 > $R/gpurun_out/final/bench_c4.log 2>&1; tail -1 $R/gpurun_out/final/bench_c4.log > $R/gpurun_out/final/bench_c4_1gpu.json
python $R/profiles/host_path.py > $R/gpurun_out/final/host_path.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/stats -- python3 $R/bench.py --steps 320 --warmup 32 --no-cpu-baseline --profile-stride 1000000 > $R/gpurun_out/final/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/stats_default -- python3 $R/bench.py > $R/gpurun_out/final/stats_default.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/stats_plain -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --lookahead 0 --profile-stride 1000000 > $R/gpurun_out/final/stats_plain.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/final/fetch -- python3 $R/bench.py --steps 48 --warmup 4 --no-cpu-baseline --profile-stride 1000000 > $R/gpurun_out/final/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/final/write -- python3 $R/bench.py --steps 48 --warmup 4 --no-cpu-baseline --profile-stride 1000000 > $R/gpurun_out/final/write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/final/fetch_plain -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --lookahead 0 --profile-stride 1000000 > $R/gpurun_out/final/fetch_plain.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/final/write_plain -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --lookahead 0 --profile-stride 1000000 > $R/gpurun_out/final/write_plain.log 2>&1
cat $R/gpurun_out/final/bench_c3.json | cut -c1-300
