#!/bin/bash
# Round-2 measurement set (run on the GPU box through gpurun): bench lines, rocprofv3 kernel stats of the default command,
# PMC traffic of the pass / plain kernels (separate --pmc passes), host-boundary latencies.  Output: gpurun_out/r02final/
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02final
rm -rf $O; mkdir -p $O
python $R/bench.py > $O/bench_c3_default.json 2> $O/bench_c3_default.err
python $R/bench.py --lookahead 16 --no-cpu-baseline > $O/bench_c3_depth16.json 2>/dev/null
python $R/bench.py --lookahead 0 --no-cpu-baseline --steps 300 > $O/bench_c3_plain.json 2>/dev/null
python $R/bench.py --step-dt 0.007 --no-secondary > $O/bench_c3_stepdt0.007.json 2>/dev/null
python $R/bench.py --scaling strong --bodies 512 --steps 96 --warmup 33 --no-secondary > $O/bench_c4_1gpu.json 2>/dev/null
python $R/profiles/host_path.py > $O/host_path.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -- python3 $R/bench.py > $O/stats_default.log 2>&1
B="python3 $R/bench.py --steps 96 --warmup 8 --no-cpu-baseline --no-secondary --profile-stride 1000000"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch32 -- $B --lookahead 32 > $O/fetch32.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write32 -- $B --lookahead 32 > $O/write32.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch16 -- $B --lookahead 16 > $O/fetch16.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write16 -- $B --lookahead 16 > $O/write16.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch0 -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --profile-stride 1000000 --lookahead 0 > $O/fetch0.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write0 -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --profile-stride 1000000 --lookahead 0 > $O/write0.log 2>&1
python3 $R/profiles/collect_r02.py $O
