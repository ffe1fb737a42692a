#!/bin/bash
# Round 6: where does the bench lose 1.4 us per step against the C++ probes?  (a) the host thread's NUMA placement relative to the GPU's PCIe
# root, (b) the library's own kernel timing (HIP-event / completion-signal sampling) being on in the bench.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06aff; rm -rf $O; mkdir -p $O
L="-I include -L hydrochrono_amd/lib -Wl,-rpath,$PWD/hydrochrono_amd/lib"
g++ -O2 -std=c++17 profiles/host_path_c.cpp $L -lhydrochrono_amd -o /tmp/hostc || exit 1
g++ -O2 -std=c++17 profiles/ahead_probe.cpp $L -lhydrochrono_amd -o /tmp/ahead || exit 1
{
echo "== topology"; nproc; lscpu | grep -E "NUMA|Model name|Socket" ; 
for d in /sys/class/drm/card*/device; do [ -e $d/vendor ] && echo "$d vendor $(cat $d/vendor) numa $(cat $d/numa_node 2>/dev/null) local_cpulist $(cat $d/local_cpulist 2>/dev/null)"; done
for d in /sys/class/kfd/kfd/topology/nodes/*; do echo "$d: $(grep -E 'simd_count|cpu_cores_count' $d/properties | tr '\n' ' ')"; done
echo "this shell may run on: $(taskset -p $$)"
} > $O/topology.txt 2>&1
CPUS=$(for d in /sys/class/drm/card*/device; do if [ "$(cat $d/vendor 2>/dev/null)" = "0x1002" ]; then cat $d/local_cpulist; break; fi; done)
echo "GPU-local cpus: $CPUS" >> $O/topology.txt
{
for rep in 1 2; do
echo "== unpinned"; /tmp/ahead 0 0 0 | cut -c1-130
if [ -n "$CPUS" ]; then echo "== taskset -c $CPUS (GPU-local)"; taskset -c $CPUS /tmp/ahead 0 0 0 | cut -c1-130; fi
FIRST=$(echo $CPUS | sed 's/[-,].*//'); if [ -n "$FIRST" ]; then echo "== taskset -c $FIRST (one GPU-local core)"; taskset -c $FIRST /tmp/ahead 0 0 0 | cut -c1-130; fi
echo "== taskset -c 0"; taskset -c 0 /tmp/ahead 0 0 0 | cut -c1-130
LAST=$(( $(nproc) - 1 )); echo "== taskset -c $LAST"; taskset -c $LAST /tmp/ahead 0 0 0 | cut -c1-130
done
} > $O/affinity_ahead_probe.txt 2>&1
{
B="python bench.py --no-secondary --no-cpu-baseline --steps 640"
echo "== bench default"; $B | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['value']), j['ms_per_step'], j['median_ms_per_step'])"
echo "== bench --profile-stride 1000000"; $B --profile-stride 1000000 | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['value']), j['ms_per_step'], j['median_ms_per_step'])"
echo "== bench --python-loop"; $B --python-loop | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['value']), j['ms_per_step'], j['median_ms_per_step'])"
if [ -n "$CPUS" ]; then echo "== bench under taskset -c $CPUS"; taskset -c $CPUS $B | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['value']), j['ms_per_step'], j['median_ms_per_step'])"; fi
echo "== bench OMP_NUM_THREADS=1 MKL_NUM_THREADS=1"; OMP_NUM_THREADS=1 MKL_NUM_THREADS=1 $B | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['value']), j['ms_per_step'], j['median_ms_per_step'])"
} > $O/bench_variants.txt 2>&1
cat $O/topology.txt $O/affinity_ahead_probe.txt $O/bench_variants.txt
