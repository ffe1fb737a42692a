#!/bin/bash
# Round 6: a caller that leaves gaps (every Chrono loop) -- is parking the queue still the right thing with the packet ring in device memory?
# HC_ARM = 0 (never park) / 1 (adaptive, default) / 2 (always), gaps of 10 ... 1000 us, C3 and one body; and the device-side gate once more.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06gaps; rm -rf $O; mkdir -p $O
L="-I include -L hydrochrono_amd/lib -Wl,-rpath,$PWD/hydrochrono_amd/lib"
g++ -O2 -std=c++17 profiles/ahead_probe.cpp $L -lhydrochrono_amd -o /tmp/ahead || exit 1
g++ -O2 -std=c++17 profiles/ahead_probe.cpp $L -lhydrochrono_amd_tuning -o /tmp/ahead_t || exit 1
g++ -O2 -std=c++17 profiles/host_path_c.cpp $L -lhydrochrono_amd -o /tmp/hostc || exit 1
{
for gap in 10 30 100 300 1000; do for arm in 0 1 2; do
  echo "== gap $gap HC_ARM=$arm"; HC_ARM=$arm timeout 120 /tmp/ahead 0 1 $gap | cut -c1-150
done; done
} > $O/arm_modes_by_gap.txt 2>&1
{
for arm in 0 1 2; do echo "== HC_ARM=$arm, 100 us of host work"; HC_ARM=$arm timeout 120 /tmp/hostc 100; done
} > $O/arm_modes_host_path.txt 2>&1
{
for gap in 30 100 300; do for gate in 0 2 0 2 0 2; do
  echo "== gap $gap HC_ARM_DEVICE_GATE=$gate"; HC_ARM_DEVICE_GATE=$gate timeout 120 /tmp/ahead_t 0 1 $gap | cut -c1-150
done; done
} > $O/device_gate_and_ab2.txt 2>&1
cat $O/arm_modes_by_gap.txt
