"""Round-5 measurement files: gpurun_out/r05/ -> profiles/r05/ (text files as they are; the bench lines of the depth-64 sweep
condensed into one table, the full JSON lines stay in gpurun_out/)."""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r05")
DST = os.path.join(ROOT, "profiles", "r05")
os.makedirs(DST, exist_ok=True)

for pat in ("*.txt", "*.csv", "pass_pmc_depth*.json", "pmc_*.json", "bench_default*.json", "bench_driver*.json", "roofline*.json"):
    for f in glob.glob(os.path.join(SRC, pat)):
        shutil.copy(f, DST)
# functional multi-context / multi-rank runs on the ONE GPU (profiles/r05_multi_functional.sh): the JSON line only
for f in glob.glob(os.path.join(SRC, "bench_c4_*_one_gpu.json")) + glob.glob(os.path.join(SRC, "bench_c4_*share_gpu*.json")):
    lines = [ln for ln in open(f) if ln.startswith("{")]
    if lines:
        open(os.path.join(DST, os.path.basename(f)), "w").write(lines[-1])
peak = os.path.join(ROOT, "gpurun_out", "r05_mfma_f64_peak.txt")
if os.path.exists(peak):
    shutil.copy(peak, os.path.join(DST, "mfma_f64_peak.txt"))

rows = []
for f in sorted(glob.glob(os.path.join(SRC, "bench_c3_depth*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:
        continue
    r, ss = d["roofline"], d.get("steady_state", {})
    rows.append((os.path.basename(f)[len("bench_c3_"):-len(".json")], d["value"], r["mean_kernel_us"], r["units_per_launch"], r["frac"],
                 r.get("fp64_frac_of_mfma_peak", 0.0), r.get("scatter_kernel_us", 0.0), r.get("step_kernel_us", 0.0),
                 ss.get("mean_ms_per_step", 0.0) * 1e3, ss.get("median_ms_per_step", 0.0) * 1e3, d.get("parity", {}).get("all_steps_max_rel_err")))
if rows:
    with open(os.path.join(DST, "depth64_sweep.txt"), "w") as fh:
        fh.write("C3 (64 bodies, S = 1024, irregular waves), python bench.py --steps 640 --warmup 8 --lookahead L on the tuning build; depth64_mtM_rR = conv_block_kernel<M, R, 4, 1>\n"
                 "(four blocks of 16 steps per streamed K word), depth64il_* = the same with the B operands formed between the MFMA groups (profiles/r05_depth64*.sh);\n"
                 "parity of the depth: tests/test_gpu_parity.py::test_depth64_experimental_pass_against_oracle (1e-11 of the oracle, both dispatch paths)\n\n")
        fh.write(f"{'variant':22s} {'evals/s':>8s} {'pass us':>8s} {'steps':>5s} {'us/step':>8s} {'of HBM':>7s} {'of MFMA':>8s} {'scatter':>8s} {'step k.':>8s} {'steady mean':>12s} {'median':>7s}\n")
        for n, v, us, L, frac, mf, sc, st, sm, smed, par in rows:
            fh.write(f"{n:22s} {v:8.0f} {us:8.1f} {L:5d} {us / L:8.2f} {frac:7.3f} {mf:8.3f} {sc:8.1f} {st:8.1f} {sm:12.2f} {smed:7.2f}\n")
print("\n".join(sorted(os.listdir(DST))))
