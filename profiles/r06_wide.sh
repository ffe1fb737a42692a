#!/bin/bash
# r06_wide.sh -- the step kernels after the LDS reads were hoisted in near_slice too: bitwise A/Bs, the wide tests, and a C4/8 rank timed
O=gpurun_out/r06wide; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_ahead.py tests/test_gpu_multi.py tests/test_chrono_adapter.py -x -q -m gpu > $O/pytest_wide.txt 2>&1; tail -4 $O/pytest_wide.txt
(echo "== pass at block start (HC_PASS_AHEAD=0)"; HC_PASS_AHEAD=0 W=8 python profiles/shard_probe.py; echo "== the default (adaptive: one block ahead on the pass lane for a wide system)"; W=8 python profiles/shard_probe.py) 2>/dev/null > $O/shard_probe_c4_rank.txt
cat $O/shard_probe_c4_rank.txt | cut -c1-200
L="-I include -L hydrochrono_amd/lib -Wl,-rpath,$PWD/hydrochrono_amd/lib"
g++ -O2 -std=c++17 profiles/ahead_probe.cpp $L -lhydrochrono_amd -o /tmp/ahead && /tmp/ahead 1 2>/dev/null > $O/ahead_probe.txt; grep "of 512" $O/ahead_probe.txt | cut -c1-130
