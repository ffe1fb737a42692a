#!/bin/bash
# Round 6, first GPU session: the synchronous step's chain at C3 taken apart.
#  (1) stage clock of finalize_kernel<4, true> (tuning build, profiles/step_stamps_probe.cpp), back to back and with 100 us gaps
#  (2) finalize_pre_kernel (kernel-argument preload: K words / scatter results requested at wave start), A/B by the same probe and by
#      ahead_probe, parity by the randomised differential run with HC_STEP_PRELOAD=1
#  (3) the AQL ring in device memory (HSA_ALLOCATE_QUEUE_DEV_MEM=1), A/B on the release library
#  (4) what the scatter costs beside a pass one block ahead: schedule x sub-block size x scatter skipped (timing bound)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06chain; rm -rf $O; mkdir -p $O
L="-I include -L hydrochrono_amd/lib -Wl,-rpath,$PWD/hydrochrono_amd/lib"
g++ -O2 -std=c++17 profiles/step_stamps_probe.cpp $L -lhydrochrono_amd_tuning -o /tmp/stamps || exit 1
g++ -O2 -std=c++17 profiles/ahead_probe.cpp $L -lhydrochrono_amd -o /tmp/ahead || exit 1
g++ -O2 -std=c++17 profiles/ahead_probe.cpp $L -lhydrochrono_amd_tuning -o /tmp/ahead_t || exit 1
g++ -O2 -std=c++17 profiles/host_path_c.cpp $L -lhydrochrono_amd -o /tmp/hostc || exit 1
T="timeout 300"
{
for rep in 1 2; do
  echo "=== stage clock, shipped step kernel, gap 0"; $T /tmp/stamps 0 0
  echo "=== stage clock, HC_STEP_PRELOAD=1, gap 0"; HC_STEP_PRELOAD=1 $T /tmp/stamps 0 0
done
echo "=== stage clock, shipped step kernel, gap 100 us"; $T /tmp/stamps 100 0
echo "=== stage clock, HC_STEP_PRELOAD=1, gap 100 us"; HC_STEP_PRELOAD=1 $T /tmp/stamps 100 0
} > $O/step_stamps.txt 2>&1
{
for rep in 1 2 3; do
  echo "== shipped"; $T /tmp/ahead_t 0 0 0
  echo "== HC_STEP_PRELOAD=1"; HC_STEP_PRELOAD=1 $T /tmp/ahead_t 0 0 0
done
echo "== shipped, gap 100"; $T /tmp/ahead_t 0 0 100
echo "== HC_STEP_PRELOAD=1, gap 100"; HC_STEP_PRELOAD=1 $T /tmp/ahead_t 0 0 100
} > $O/preload_ab.txt 2>&1
{
for rep in 1 2 3; do
  echo "== default ring (host memory)"; $T /tmp/ahead 0 0 0
  echo "== HSA_ALLOCATE_QUEUE_DEV_MEM=1"; HSA_ALLOCATE_QUEUE_DEV_MEM=1 $T /tmp/ahead 0 0 0
done
echo "== default ring, gap 100"; $T /tmp/ahead 0 0 100
echo "== HSA_ALLOCATE_QUEUE_DEV_MEM=1, gap 100"; HSA_ALLOCATE_QUEUE_DEV_MEM=1 $T /tmp/ahead 0 0 100
echo "== host_path_c default ring"; $T /tmp/hostc
echo "== host_path_c HSA_ALLOCATE_QUEUE_DEV_MEM=1"; HSA_ALLOCATE_QUEUE_DEV_MEM=1 $T /tmp/hostc
echo "=== stage clock with the ring in device memory, gap 0"; HSA_ALLOCATE_QUEUE_DEV_MEM=1 $T /tmp/stamps 0 0
echo "=== stage clock with the ring in device memory + preload, gap 0"; HSA_ALLOCATE_QUEUE_DEV_MEM=1 HC_STEP_PRELOAD=1 $T /tmp/stamps 0 0
} > $O/queue_dev_mem_ab.txt 2>&1
{
for sched in 0 1; do for sb in 0 8; do for skip in 0 1; do
  echo "== schedule $sched  HC_SUB_BLOCK=$sb  HC_SKIP_SCATTER=$skip"
  BY_POSITION=1 HC_SUB_BLOCK=$sb HC_SKIP_SCATTER=$skip $T /tmp/ahead_t 0 $sched 0
done; done; done
} > $O/scatter_matrix.txt 2>&1
HC_STEP_PRELOAD=1 timeout 200 python profiles/fuzz_parity.py 90 610001 > $O/fuzz_preload.txt 2>&1
tail -3 $O/fuzz_preload.txt
cat $O/step_stamps.txt | head -80
