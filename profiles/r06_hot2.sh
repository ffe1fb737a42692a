#!/bin/bash
# Round 6, third GPU session: step_hot_kernel with the tables requested first (hydrostatic / wave terms formed in the shadow of the K
# words), the device-side gate of a parked queue, and the pass lane's share of the chip under "one block ahead" at gap 0.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06hot2; rm -rf $O; mkdir -p $O
L="-I include -L hydrochrono_amd/lib -Wl,-rpath,$PWD/hydrochrono_amd/lib"
g++ -O2 -std=c++17 profiles/step_stamps_probe.cpp $L -lhydrochrono_amd_tuning -o /tmp/stamps || exit 1
g++ -O2 -std=c++17 profiles/ahead_probe.cpp $L -lhydrochrono_amd -o /tmp/ahead || exit 1
g++ -O2 -std=c++17 profiles/ahead_probe.cpp $L -lhydrochrono_amd_tuning -o /tmp/ahead_t || exit 1
g++ -O2 -std=c++17 profiles/host_path_c.cpp $L -lhydrochrono_amd -o /tmp/hostc || exit 1
T="timeout 300"
timeout 900 python -m pytest tests/test_gpu_boundary.py -x -q -m gpu > $O/pytest_boundary.txt 2>&1
tail -3 $O/pytest_boundary.txt
{
for rep in 1 2; do
  echo "=== stage clock, step_hot_kernel, gap 0"; $T /tmp/stamps 0 0
done
echo "=== stage clock, step_hot_kernel, gap 100 us, device-side gate (default)"; $T /tmp/stamps 100 0
echo "=== stage clock, step_hot_kernel, gap 100 us, HC_ARM_DEVICE_GATE=0"; HC_ARM_DEVICE_GATE=0 $T /tmp/stamps 100 0
echo "=== stage clock, step_hot_kernel, gap 100 us, device-side gate (default)"; $T /tmp/stamps 100 0
echo "=== stage clock, step_hot_kernel, gap 100 us, HC_ARM_DEVICE_GATE=0"; HC_ARM_DEVICE_GATE=0 $T /tmp/stamps 100 0
} > $O/step_stamps.txt 2>&1
{
for gap in 0 30 100 300; do for gate in 1 0; do
  echo "== gap $gap HC_ARM_DEVICE_GATE=$gate"; HC_ARM_DEVICE_GATE=$gate $T /tmp/ahead_t 0 0 $gap;  HC_ARM_DEVICE_GATE=$gate $T /tmp/ahead_t 0 1 $gap
done; done
} > $O/device_gate_ab.txt 2>&1
{
for free in 4 8 12 16 20 24; do for sl in 0 8 16; do
  echo "== HC_PASS_FREE_CUS=$free HC_PASS_SLICES=$sl"; BY_POSITION=1 HC_PASS_FREE_CUS=$free HC_PASS_SLICES=$sl $T /tmp/ahead_t 0 1 0
done; done
} > $O/pass_lane_share.txt 2>&1
{ echo "== release"; $T /tmp/hostc; $T /tmp/hostc 100; $T /tmp/ahead 0; } > $O/release_probes.txt 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt
grep -E "median|doorbell ->|===" $O/step_stamps.txt | cut -c1-140
