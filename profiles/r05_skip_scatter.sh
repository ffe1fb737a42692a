cd $GRAFT_REPO_ROOT
g++ -O2 -std=c++17 profiles/ahead_probe.cpp -I include -L hydrochrono_amd/lib -lhydrochrono_amd_tuning -Wl,-rpath,$PWD/hydrochrono_amd/lib -o /tmp/ahead_probe_t || exit 1
for rep in 1 2; do
echo "== with scatter"; /tmp/ahead_probe_t 0 0 0 2>/dev/null
echo "== HC_SKIP_SCATTER=1 (timing bound only)"; HC_SKIP_SCATTER=1 /tmp/ahead_probe_t 0 0 0 2>/dev/null
done
