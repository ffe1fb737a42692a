#!/bin/bash
# round 5: the adaptive pass schedule -- GPU tests of the rule, then the crossover between the two schedules by the caller's gap
# (profiles/ahead_probe.cpp with FINE_GAPS: both schedules and the adaptive default at gaps of 0 .. 20 us, C3 and a C4/8 rank)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_ahead.py -x -q -m gpu -k "adaptive" -s > $O/test_adaptive.log 2>&1; echo "adaptive tests rc=$?"
tail -5 $O/test_adaptive.log
g++ -O2 -std=c++17 profiles/ahead_probe.cpp -I include -L hydrochrono_amd/lib -lhydrochrono_amd -Wl,-rpath,$R/hydrochrono_amd/lib -o /tmp/ahead_probe || exit 1
FINE_GAPS=1 /tmp/ahead_probe 1 2>/dev/null > $O/ahead_probe_fine_gaps.txt
cat $O/ahead_probe_fine_gaps.txt
