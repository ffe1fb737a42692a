#!/bin/bash
# round 5: the depth-64 pass (conv_block_kernel<MT, R, 4>: four blocks of 16 steps per streamed K word) against depth 32 at C3 --
# parity through the suite's test, then bench lines per (MT, R) variant: pass time per launch and per step, steady-state step latency
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "depth64" > $O/test_depth64.log 2>&1; echo "depth64 tests rc=$?"; tail -3 $O/test_depth64.log
B="python bench.py --steps 640 --warmup 8 --no-cpu-baseline --no-c4-share --no-c4-one-gpu"
$B --lookahead 32 > $O/bench_c3_depth32_ref.json 2>/dev/null
for mt in 3 4 6; do for r in 3 4 5; do
  HC_BLOCK64_MT=$mt HC_BLOCK64_R=$r $B --lookahead 64 > $O/bench_c3_depth64_mt${mt}_r${r}.json 2>/dev/null
done; done
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r05")
for f in sorted(glob.glob(O + "/bench_c3_depth*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    r, ss = d["roofline"], d.get("steady_state", {})
    print(f"{os.path.basename(f):38s} value {d['value']:8.0f}  ms/step {d['ms_per_step']:.5f}  pass {r['mean_kernel_us']:7.1f} us / {r['units_per_launch']} steps = {r['mean_kernel_us'] / r['units_per_launch']:.2f} us/step  frac {r['frac']:.3f}  fp64 {r.get('fp64_frac_of_mfma_peak', 0):.3f}  scatter {r.get('scatter_kernel_us', 0):.1f}  step {r.get('step_kernel_us', 0):.1f}  steady mean {ss.get('mean_ms_per_step', 0) * 1e3:.2f} median {ss.get('median_ms_per_step', 0) * 1e3:.2f}  parity {d.get('parity_max_rel_err_vs_oracle')}")
PY
