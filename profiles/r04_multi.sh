#!/bin/bash
# round 4: the fan-out of hc_step_multi (worker thread per context) against the one-thread form, C3 and C4 arrays in G contexts on the one GPU
set -x
mkdir -p gpurun_out/r04
g++ -O2 -std=c++17 profiles/multi_path_c.cpp -I include -L hydrochrono_amd/lib -lhydrochrono_amd -Wl,-rpath,$PWD/hydrochrono_amd/lib -pthread -o /tmp/multi_path_c || exit 1
(echo "== worker thread per context (default)"; /tmp/multi_path_c 64 1024 3000; echo "== one thread (HC_MULTI_THREADS=0)"; HC_MULTI_THREADS=0 /tmp/multi_path_c 64 1024 3000) > gpurun_out/r04/multi_path_c.txt 2>&1
(echo "== worker thread per context (default)"; /tmp/multi_path_c 512 1024 600; echo "== one thread (HC_MULTI_THREADS=0)"; HC_MULTI_THREADS=0 /tmp/multi_path_c 512 1024 600) > gpurun_out/r04/multi_path_c_c4.txt 2>&1
cat gpurun_out/r04/multi_path_c.txt gpurun_out/r04/multi_path_c_c4.txt
