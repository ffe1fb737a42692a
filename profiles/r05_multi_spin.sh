#!/bin/bash
# round 5: what the spin time of the fan-out's worker threads (HC_MULTI_SPIN_US) costs and buys -- hc_step_multi over 4 contexts of the
# C3 array on the one GPU, with 0 / 100 / 300 / 1000 us of host work between the calls, spin times 0 / 50 / 200 / 1000 us.
# (While a worker spins it holds a core; once it sleeps, the next call wakes it through the kernel.)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
g++ -O2 -std=c++17 profiles/multi_path_c.cpp -I include -L hydrochrono_amd/lib -lhydrochrono_amd -Wl,-rpath,$R/hydrochrono_amd/lib -pthread -o /tmp/multi_path_c || exit 1
for gap in 0 100 300 1000; do for spin in 0 50 200 1000; do
  echo "== host work between calls ${gap} us, HC_MULTI_SPIN_US=${spin}"
  GAP_US=$gap ONLY_G=4 HC_MULTI_SPIN_US=$spin /tmp/multi_path_c 64 1024 1500 2>/dev/null | grep "G = 4"
done; done > $O/multi_spin.txt
cat $O/multi_spin.txt
