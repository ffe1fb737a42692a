#!/bin/bash
# r06_hdp.sh -- what does the HDP write-back in front of EVERY doorbell cost (packet ring in host memory)?
O=gpurun_out/r06hdp; mkdir -p $O
L="-I include -L hydrochrono_amd/lib -Wl,-rpath,$PWD/hydrochrono_amd/lib"
g++ -O2 -std=c++17 profiles/host_path_c.cpp $L -lhydrochrono_amd -o /tmp/host_path_c && {
  for i in 1 2 3; do echo "== HC_HDP_FLUSH=0"; HC_HDP_FLUSH=0 /tmp/host_path_c; echo "== HC_HDP_FLUSH=1"; HC_HDP_FLUSH=1 /tmp/host_path_c; done
  echo "== 100 us gaps, HC_HDP_FLUSH=0"; HC_HDP_FLUSH=0 /tmp/host_path_c 100; echo "== 100 us gaps, HC_HDP_FLUSH=1"; HC_HDP_FLUSH=1 /tmp/host_path_c 100; } > $O/host_path_c_hdp_flush.txt 2>&1
cut -c1-150 $O/host_path_c_hdp_flush.txt
for f in 0 1; do HC_HDP_FLUSH=$f python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('HC_HDP_FLUSH=$f driver command: value', round(j['value']), 'median', round(j['median_ms_per_step']*1e3, 2))
"; done | tee $O/bench_hdp_flush.txt
