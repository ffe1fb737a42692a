"""Round-6 additions to the committed summaries (after profiles/collect_r04.py <dir> r06): the HBM traffic of ONE steady-state step from the
separate --pmc passes (profiles/r06/pmc_step_traffic.json: what bench.py's roofline.whole_step cites), and the functional multi-context /
multi-rank lines reduced to their JSON line.   Usage: python profiles/collect_r06.py [gpurun_out/r06final]"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r06final")
DST = os.path.join(ROOT, "profiles", "r06")
os.makedirs(DST, exist_ok=True)

for name in os.listdir(SRC):
    if name.startswith("bench_c4_") and name.endswith(".json"):
        lines = [ln for ln in open(os.path.join(SRC, name)) if ln.startswith("{")]
        if lines:
            open(os.path.join(DST, name), "w").write(lines[-1])
if os.path.exists(os.path.join(SRC, "pytest_gpu.txt")):
    tail = open(os.path.join(SRC, "pytest_gpu.txt")).read().strip().splitlines()[-3:]
    open(os.path.join(DST, "pytest_gpu_tail.txt"), "w").write("\n".join(tail) + "\n")

p = os.path.join(DST, "pmc_traffic.json")
if os.path.exists(p):
    pmc = json.load(open(p))
    fetch, write = pmc.get("fetch32", {}), pmc.get("write32", {})

    def bytes_of(prefix, grid=None):
        """mean HBM bytes per dispatch of the kernels whose name starts with `prefix` (FETCH x 2 for wide streaming reads + WRITE), and the dispatches"""
        tot, n = 0.0, 0
        for k, v in fetch.items():
            if k.startswith(prefix) and (grid is None or k.endswith(f"grid={grid}")):
                w = write.get(k, {"mean_KB": 0.0})
                tot += v["dispatches"] * 1024.0 * (2.0 * v["mean_KB"] + w["mean_KB"])
                n += v["dispatches"]
        return (tot / n if n else 0.0), n

    lookahead = 32
    pass_b, pass_n = bytes_of("hc::conv_block_kernel")
    red_b, red_n = bytes_of("hc::reduce_block_kernel")
    scat_b, scat_n = bytes_of("hc::scatter_kernel")
    hot_b, hot_n = bytes_of("hc::step_hot_kernel")
    gen_b, gen_n = bytes_of("hc::finalize_kernel")
    step_b = hot_b if hot_n else gen_b
    per_step = pass_b / lookahead + red_b / lookahead + scat_b + step_b
    out = {"hbm_bytes_per_steady_state_step": per_step,
           "parts": {"pass_per_launch": pass_b, "pass_launches": pass_n, "pass_share_per_step": pass_b / lookahead,
                     "reduce_per_launch": red_b, "reduce_share_per_step": red_b / lookahead,
                     "scatter_mean_per_launch": scat_b, "scatter_launches": scat_n,
                     "step_kernel_mean_per_launch": step_b, "step_kernel_launches": hot_n or gen_n,
                     "step_kernel": "hc::step_hot_kernel" if hot_n else "hc::finalize_kernel"},
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE runs of `python3 bench.py --steps 96 --warmup 8 --no-cpu-baseline --no-secondary "
                     "--profile-stride 1000000` (profiles/r06_final.sh); per dispatch: 1024 x (2 x FETCH_SIZE + WRITE_SIZE) KB (gfx950 correction for wide "
                     "streaming reads, MI355X_MICROARCH.md); per step: pass / 32 + reduction / 32 + mean scatter launch + mean step kernel launch",
           "source": "profiles/r06/pmc_traffic.json"}
    json.dump(out, open(os.path.join(DST, "pmc_step_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))
print("\n".join(sorted(os.listdir(DST))))
