#!/bin/bash
# Round-6 measurement set (run on the GPU box through gpurun): bench lines (C3 default with the `init` block / the driver's fixed command /
# depth 16 / off-grid step / C4 on one GPU / `--gpus 2` WITHOUT a launcher: hc_step_multi + the RCCL child ranks, functional on one GPU),
# the launcher form with both exchanges (functional), the schedules by caller gap, the host boundary, the stage clock of the step kernel,
# one C4/8 rank's share, rocprofv3 kernel stats of the DRIVER COMMAND WITHOUT ITS WIDE AND SMALL SECONDARIES and FETCH_SIZE / WRITE_SIZE
# passes (separate --pmc runs), the suite.   Output: gpurun_out/r06final/ -> profiles/collect_r04.py ... r06 + profiles/collect_r06.py
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06final
rm -rf $O; mkdir -p $O
cd $R
L="-I include -L hydrochrono_amd/lib -Wl,-rpath,$R/hydrochrono_amd/lib"
# the differential run on the library as shipped (packet rings in host memory; the fan-out's workers bound to their GPUs' CPUs)
FUZZ_SHARDS=1 timeout 300 python profiles/fuzz_parity.py 150 710001 > $O/fuzz_parity_shards_workers_bound.txt 2>&1; echo "fuzz shards rc=$?"
timeout 200 python profiles/fuzz_parity.py 90 720001 > $O/fuzz_parity_2.txt 2>&1; echo "fuzz rc=$?"
python bench.py > $O/bench_c3_default.json 2> $O/bench_c3_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c3_driver_cmd.json 2>/dev/null
python bench.py --no-pin --no-secondary --no-cpu-baseline > $O/bench_c3_default_thread_not_bound.json 2>/dev/null   # wherever the scheduler puts the stepping thread
python bench.py --no-pin --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $O/bench_c3_driver_cmd_thread_not_bound.json 2>/dev/null
HC_QUEUE_DEV_MEM=1 python bench.py --no-secondary --no-cpu-baseline > $O/bench_c3_default_dev_mem_rings.json 2>/dev/null   # the opt-in: packet rings in device memory
HC_QUEUE_DEV_MEM=1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $O/bench_c3_driver_cmd_dev_mem_rings.json 2>/dev/null
python bench.py --lookahead 16 --no-cpu-baseline --no-c4-share --no-c4-one-gpu --no-small-configs --no-init > $O/bench_c3_depth16.json 2>/dev/null
python bench.py --step-dt 0.007 --no-secondary > $O/bench_c3_stepdt0.007.json 2>/dev/null
python bench.py --scaling strong --bodies 512 --steps 256 --warmup 104 --no-secondary > $O/bench_c4_1gpu.json 2>/dev/null
HC_BENCH_CHILD_TIMEOUT_S=900 timeout 1500 python bench.py --gpus 2 --steps 64 --warmup 8 > $O/bench_c4_no_launcher_2ctx_one_gpu.json 2> $O/no_launcher.err
for ex in host rccl; do
  HC_BENCH_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2962$([ $ex = host ] && echo 1 || echo 2) bench.py --gpus 2 --steps 40 --warmup 8 --exchange $ex > $O/bench_c4_2ranks_share_gpu_$ex.json 2> $O/r2_$ex.err
done
g++ -O2 -std=c++17 profiles/ahead_probe.cpp $L -lhydrochrono_amd -o /tmp/ahead && { /tmp/ahead 1 2>/dev/null > $O/ahead_probe.txt; FINE_GAPS=1 /tmp/ahead 0 2>/dev/null > $O/ahead_probe_fine_gaps.txt; HC_QUEUE_DEV_MEM=1 /tmp/ahead 0 2>/dev/null > $O/ahead_probe_dev_mem_rings.txt; }
g++ -O2 -std=c++17 profiles/host_path_c.cpp $L -lhydrochrono_amd -o /tmp/host_path_c && { /tmp/host_path_c; /tmp/host_path_c 100; echo "== BIND=0 (stepping thread not bound to the GPU's CPUs)"; BIND=0 /tmp/host_path_c; echo "== HC_QUEUE_DEV_MEM=1 (opt-in: packet rings in device memory)"; HC_QUEUE_DEV_MEM=1 /tmp/host_path_c; HC_QUEUE_DEV_MEM=1 /tmp/host_path_c 100; } 2>/dev/null > $O/host_path_c.txt
g++ -O2 -std=c++17 profiles/step_stamps_probe.cpp $L -lhydrochrono_amd_tuning -o /tmp/stamps && {
  echo "=== step_hot_kernel (what ships), gap 0"; /tmp/stamps 0 0; echo "=== step_hot_kernel, gap 0, second run"; /tmp/stamps 0 0
  echo "=== general step kernel finalize_kernel<4, true> (HC_STEP_HOT=0), gap 0"; HC_STEP_HOT=0 /tmp/stamps 0 0
  echo "=== step_hot_kernel, stepping thread NOT bound (BIND=0), gap 0"; BIND=0 /tmp/stamps 0 0
  echo "=== step_hot_kernel, packet ring in DEVICE memory (opt-in HC_QUEUE_DEV_MEM=1), gap 0"; HC_QUEUE_DEV_MEM=1 /tmp/stamps 0 0
  echo "=== general step kernel, packet ring in DEVICE memory, gap 0"; HC_QUEUE_DEV_MEM=1 HC_STEP_HOT=0 /tmp/stamps 0 0
  echo "=== step_hot_kernel, 100 us of host work between calls"; /tmp/stamps 100 0
  echo "=== step_hot_kernel, two workgroups per row tile (tuning experiment HC_STEP_HALVES=2), gap 0"; HC_STEP_HALVES=2 /tmp/stamps 0 0; } > $O/step_stage_clock.txt 2>&1
g++ -O2 -std=c++17 profiles/multi_path_c.cpp $L -lhydrochrono_amd -pthread -o /tmp/multi_path_c && {
  (echo "== worker thread per context (default)"; /tmp/multi_path_c 512 1024 600; echo "== one thread (HC_MULTI_THREADS=0)"; HC_MULTI_THREADS=0 /tmp/multi_path_c 512 1024 600) > $O/multi_path_c_c4.txt 2>&1
}
(echo "== pass at block start (HC_PASS_AHEAD=0)"; HC_PASS_AHEAD=0 W=8 python profiles/shard_probe.py; echo "== the default (adaptive: one block ahead on the pass lane for a wide system)"; W=8 python profiles/shard_probe.py) 2>/dev/null > $O/shard_probe_c4_rank.txt
python profiles/shard_curve.py > $O/shard_curve.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-c4-share --no-c4-one-gpu --no-small-configs --no-init > $O/stats_default.log 2>&1
export W=8 HC_PASS_AHEAD=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c4rank -- python3 $R/profiles/shard_probe.py > $O/stats_c4rank.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_c4rank -- python3 $R/profiles/shard_probe.py > $O/fetch_c4rank.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write_c4rank -- python3 $R/profiles/shard_probe.py > $O/write_c4rank.log 2>&1
unset W HC_PASS_AHEAD
B="python3 $R/bench.py --steps 96 --warmup 8 --no-cpu-baseline --no-secondary --profile-stride 1000000"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch32 -- $B --lookahead 32 > $O/fetch32.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write32 -- $B --lookahead 32 > $O/write32.log 2>&1
cd $R
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
python3 profiles/collect_r04.py $O r06 > $O/collect.log 2>&1; tail -30 $O/collect.log | head -40
python3 profiles/collect_r06.py $O
