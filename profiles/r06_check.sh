#!/bin/bash
# Round 6, fourth GPU session: the suite on the new step path, two timing experiments (no acquire fence on the step kernel's packet; the
# parked queue's gate as a device-side record alone), the bench lines (N = 1 with the init block; --gpus 2 without a launcher on one GPU).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06check; rm -rf $O; mkdir -p $O
L="-I include -L hydrochrono_amd/lib -Wl,-rpath,$PWD/hydrochrono_amd/lib"
g++ -O2 -std=c++17 profiles/step_stamps_probe.cpp $L -lhydrochrono_amd_tuning -o /tmp/stamps || exit 1
g++ -O2 -std=c++17 profiles/ahead_probe.cpp $L -lhydrochrono_amd_tuning -o /tmp/ahead_t || exit 1
T="timeout 300"
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -4 $O/pytest_gpu.txt
{
for rep in 1 2 3; do
  echo "== acquire fence (shipped)"; $T /tmp/ahead_t 0 0 0
  echo "== HC_STEP_NO_ACQUIRE=1"; HC_STEP_NO_ACQUIRE=1 $T /tmp/ahead_t 0 0 0
done
echo "=== stage clock, HC_STEP_NO_ACQUIRE=1"; HC_STEP_NO_ACQUIRE=1 $T /tmp/stamps 0 0
} > $O/no_acquire_ab.txt 2>&1
{
for gap in 30 100; do for gate in 0 2 0 2; do
  echo "== gap $gap HC_ARM_DEVICE_GATE=$gate"; HC_ARM_DEVICE_GATE=$gate timeout 120 /tmp/ahead_t 0 0 $gap
done; done
} > $O/device_gate_and_ab.txt 2>&1
python bench.py > $O/bench_c3_default.json 2> $O/bench_c3_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c3_driver_cmd.json 2>/dev/null
HC_BENCH_CHILD_TIMEOUT_S=900 timeout 1500 python bench.py --gpus 2 --steps 64 --warmup 8 > $O/bench_c4_no_launcher_2ctx_one_gpu.json 2> $O/bench_c4_no_launcher.err
tail -c 600 $O/bench_c4_no_launcher.err
grep -E "median" $O/no_acquire_ab.txt | cut -c1-120
