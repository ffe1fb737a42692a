#!/bin/bash
# round 5: depth-64 pass with the B operands formed in the shadow of the MFMAs (interleaved groups), (MT, R) variants at C3
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
B="python bench.py --steps 640 --warmup 8 --no-cpu-baseline --no-c4-share --no-c4-one-gpu"
for mt in 3 6; do for r in 3 4; do
  HC_BLOCK64_MT=$mt HC_BLOCK64_R=$r $B --lookahead 64 > $O/bench_c3_depth64il_mt${mt}_r${r}.json 2>/dev/null
done; done
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r05")
for f in sorted(glob.glob(O + "/bench_c3_depth64il*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    r, ss = d["roofline"], d.get("steady_state", {})
    print(f"{os.path.basename(f):38s} value {d['value']:8.0f}  pass {r['mean_kernel_us']:7.1f} us / {r['units_per_launch']} steps = {r['mean_kernel_us'] / r['units_per_launch']:.2f} us/step  frac {r['frac']:.3f}  fp64 {r.get('fp64_frac_of_mfma_peak', 0):.3f}  steady mean {ss.get('mean_ms_per_step', 0) * 1e3:.2f} median {ss.get('median_ms_per_step', 0) * 1e3:.2f}  parity {d.get('parity', {}).get('all_steps_max_rel_err')}")
PY
