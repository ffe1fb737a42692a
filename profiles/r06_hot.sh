#!/bin/bash
# Round 6, second GPU session: the step kernel of the common block step (step_hot_kernel) and the packet ring in device memory.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06hot; rm -rf $O; mkdir -p $O
L="-I include -L hydrochrono_amd/lib -Wl,-rpath,$PWD/hydrochrono_amd/lib"
g++ -O2 -std=c++17 profiles/step_stamps_probe.cpp $L -lhydrochrono_amd_tuning -o /tmp/stamps || exit 1
g++ -O2 -std=c++17 profiles/ahead_probe.cpp $L -lhydrochrono_amd -o /tmp/ahead || exit 1
g++ -O2 -std=c++17 profiles/ahead_probe.cpp $L -lhydrochrono_amd_tuning -o /tmp/ahead_t || exit 1
g++ -O2 -std=c++17 profiles/host_path_c.cpp $L -lhydrochrono_amd -o /tmp/hostc || exit 1
T="timeout 300"
timeout 900 python -m pytest tests/test_gpu_boundary.py -x -q -m gpu > $O/pytest_boundary.txt 2>&1
tail -5 $O/pytest_boundary.txt
{
for rep in 1 2; do
  echo "=== stage clock, step_hot_kernel (default), gap 0"; $T /tmp/stamps 0 0
  echo "=== stage clock, general step kernel (HC_STEP_HOT=0), gap 0"; HC_STEP_HOT=0 $T /tmp/stamps 0 0
done
echo "=== stage clock, step_hot_kernel, ring in HOST memory (HC_QUEUE_DEV_MEM=0), gap 0"; HC_QUEUE_DEV_MEM=0 $T /tmp/stamps 0 0
echo "=== stage clock, step_hot_kernel, gap 100 us"; $T /tmp/stamps 100 0
echo "=== stage clock, general step kernel, gap 100 us"; HC_STEP_HOT=0 $T /tmp/stamps 100 0
} > $O/step_stamps.txt 2>&1
{
for rep in 1 2 3; do
  echo "== hot"; $T /tmp/ahead_t 0 0 0
  echo "== HC_STEP_HOT=0"; HC_STEP_HOT=0 $T /tmp/ahead_t 0 0 0
done
} > $O/hot_ab.txt 2>&1
{
echo "== release library, defaults"; $T /tmp/ahead 0
echo "== release library, HC_QUEUE_DEV_MEM=0"; HC_QUEUE_DEV_MEM=0 $T /tmp/ahead 0
echo "== fine gaps"; FINE_GAPS=1 $T /tmp/ahead 0
} > $O/ahead_probe.txt 2>&1
{ echo "== defaults"; $T /tmp/hostc; echo "== HC_QUEUE_DEV_MEM=0"; HC_QUEUE_DEV_MEM=0 $T /tmp/hostc; } > $O/host_path_c.txt 2>&1
python bench.py --no-c4-share --no-c4-one-gpu --no-small-configs > $O/bench_c3_default.json 2> $O/bench_c3_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-c4-share --no-c4-one-gpu --no-small-configs > $O/bench_c3_driver_cmd.json 2>/dev/null
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt
cat $O/hot_ab.txt | cut -c1-140
