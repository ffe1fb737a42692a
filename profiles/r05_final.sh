#!/bin/bash
# Round-5 measurement set (run on the GPU box through gpurun): bench lines (C3 default / the driver's fixed command / C4 on one GPU),
# the schedules by caller gap (ahead_probe: pinned 0 / 1 and the adaptive default), the host boundary, one C4/8 rank's share,
# rocprofv3 kernel stats of the DRIVER COMMAND WITHOUT ITS WIDE AND SMALL SECONDARIES (so that the pass kernel's 256-workgroup row holds
# C3 launches only: the file bench.py names in roofline.profile_file) and FETCH_SIZE / WRITE_SIZE passes (separate --pmc runs).
# Output: gpurun_out/r05final/ -> profiles/collect_r04.py gpurun_out/r05final r05 -> profiles/r05/
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05final
rm -rf $O; mkdir -p $O
python $R/bench.py > $O/bench_c3_default.json 2> $O/bench_c3_default.err
python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c3_driver_cmd.json 2>/dev/null
python $R/bench.py --lookahead 16 --no-cpu-baseline --no-c4-share --no-c4-one-gpu --no-small-configs > $O/bench_c3_depth16.json 2>/dev/null
python $R/bench.py --step-dt 0.007 --no-secondary > $O/bench_c3_stepdt0.007.json 2>/dev/null
python $R/bench.py --scaling strong --bodies 512 --steps 256 --warmup 104 --no-secondary > $O/bench_c4_1gpu.json 2>/dev/null
g++ -O2 -std=c++17 $R/profiles/ahead_probe.cpp -I $R/include -L $R/hydrochrono_amd/lib -lhydrochrono_amd -Wl,-rpath,$R/hydrochrono_amd/lib -o /tmp/ahead_probe && /tmp/ahead_probe 1 2>/dev/null > $O/ahead_probe.txt
g++ -O2 -std=c++17 $R/profiles/host_path_c.cpp -I $R/include -L $R/hydrochrono_amd/lib -lhydrochrono_amd -Wl,-rpath,$R/hydrochrono_amd/lib -o /tmp/host_path_c && /tmp/host_path_c 2>/dev/null > $O/host_path_c.txt
g++ -O2 -std=c++17 $R/profiles/multi_path_c.cpp -I $R/include -L $R/hydrochrono_amd/lib -lhydrochrono_amd -Wl,-rpath,$R/hydrochrono_amd/lib -pthread -o /tmp/multi_path_c && {
  (echo "== worker thread per context (default)"; /tmp/multi_path_c 512 1024 600; echo "== one thread (HC_MULTI_THREADS=0)"; HC_MULTI_THREADS=0 /tmp/multi_path_c 512 1024 600) > $O/multi_path_c_c4.txt 2>&1
}
(echo "== pass at block start (HC_PASS_AHEAD=0)"; HC_PASS_AHEAD=0 W=8 python $R/profiles/shard_probe.py; echo "== the default (adaptive: one block ahead on the pass lane for a wide system)"; W=8 python $R/profiles/shard_probe.py) 2>/dev/null > $O/shard_probe_c4_rank.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-c4-share --no-c4-one-gpu --no-small-configs > $O/stats_default.log 2>&1
export W=8 HC_PASS_AHEAD=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c4rank -- python3 $R/profiles/shard_probe.py > $O/stats_c4rank.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_c4rank -- python3 $R/profiles/shard_probe.py > $O/fetch_c4rank.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write_c4rank -- python3 $R/profiles/shard_probe.py > $O/write_c4rank.log 2>&1
unset W HC_PASS_AHEAD
B="python3 $R/bench.py --steps 96 --warmup 8 --no-cpu-baseline --no-secondary --profile-stride 1000000"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch32 -- $B --lookahead 32 > $O/fetch32.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write32 -- $B --lookahead 32 > $O/write32.log 2>&1
python3 $R/profiles/collect_r04.py $O r05
