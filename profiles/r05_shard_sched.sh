#!/bin/bash
# round 5: the two pass schedules for row shards of C4 larger than a C4/8 rank (128 / 256 / 512 bodies of 512: K slices of 19 / 39 / 77 GB)
# by the caller's gap, C++ caller -- the data behind the adaptive rule's threshold for wide systems.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
g++ -O2 -std=c++17 profiles/ahead_probe.cpp -I include -L hydrochrono_amd/lib -lhydrochrono_amd -Wl,-rpath,$R/hydrochrono_amd/lib -o /tmp/ahead_probe || exit 1
for rows in 64 128 256 512; do SHARD_ROWS=$rows /tmp/ahead_probe 1 2>/dev/null; done > $O/ahead_probe_shard_sizes.txt
cut -c1-150 $O/ahead_probe_shard_sizes.txt
