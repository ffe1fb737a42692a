"""Turns the raw output of profiles/r04_final.sh / r05_final.sh (gpurun_out/r0Nfinal/) into the committed summaries under profiles/r0N/.
Usage: python profiles/collect_r04.py [gpurun_out/r04final] [r04]"""
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r04final")
ROUND = sys.argv[2] if len(sys.argv) > 2 else "r04"
DST = os.path.join(ROOT, "profiles", ROUND)
os.makedirs(DST, exist_ok=True)


def short(name):
    return re.sub(r"\(.*", "", name).replace("void ", "").strip()


def stats_file(subdir):
    for path in sorted(glob.glob(os.path.join(SRC, subdir, "*", "*_kernel_stats.csv")), key=os.path.getmtime, reverse=True):
        if "hc::" in open(path).read():
            return path
    return None


def counter_by_kernel(subdir, counter):
    """mean counter value per (kernel, grid size): the short passes of the two-level form run the pass kernel on a small grid"""
    out = {}
    for path in sorted(glob.glob(os.path.join(SRC, subdir, "*", "*_counter_collection.csv")), key=os.path.getmtime):
        acc = {}
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter or "hc::" not in row["Kernel_Name"]:
                    continue
                grid = row.get("Grid_Size") or row.get("Grid_Size_X") or "?"
                acc.setdefault(f"{short(row['Kernel_Name'])} grid={grid}", []).append(float(row["Counter_Value"]))
        if acc:
            out = {k: {"dispatches": len(v), "mean_KB": sum(v) / len(v)} for k, v in sorted(acc.items())}
    return out


for sub, name in (("stats_default", "c3_driver_cmd_kernel_stats.csv"), ("stats_c4rank", "c4_rank_share_kernel_stats.csv")):
    p = stats_file(sub)
    if p:
        shutil.copy(p, os.path.join(DST, name))
for name in sorted(os.listdir(SRC)):
    p = os.path.join(SRC, name)
    if os.path.isfile(p) and os.path.getsize(p) > 0 and (name.endswith(".txt") or (name.endswith(".json") and name.startswith(("bench_", "host_path")))):
        if name.endswith(".json") and name.startswith("bench_"):
            lines = [ln for ln in open(p) if ln.startswith("{")]
            if not lines:
                continue
            open(os.path.join(DST, name), "w").write(lines[-1])
        else:
            shutil.copy(p, os.path.join(DST, name))
lines = [ln for ln in open(os.path.join(SRC, "stats_default.log")) if ln.startswith("{")] if os.path.exists(os.path.join(SRC, "stats_default.log")) else []
if lines:
    open(os.path.join(DST, "bench_c3_driver_cmd_under_rocprof.json"), "w").write(lines[-1])
pmc = {k: counter_by_kernel(k, "FETCH_SIZE" if k.startswith("fetch") else "WRITE_SIZE") for k in ("fetch_c4rank", "write_c4rank", "fetch32", "write32")}
pmc["units"] = "KB as reported by rocprofv3 (raw); gfx950 correction for wide streaming reads: FETCH_SIZE x2 (MI355X_MICROARCH.md, HBM)"
json.dump(pmc, open(os.path.join(DST, "pmc_traffic.json"), "w"), indent=1)
# C3 pass traffic for bench.py's roofline.traffic (per launch)
f32 = next((v for k, v in pmc["fetch32"].items() if k.startswith("hc::conv_block_kernel")), None)
w32 = next((v for k, v in pmc["write32"].items() if k.startswith("hc::conv_block_kernel")), None)
if f32 and w32:
    tpath = os.path.join(ROOT, "profiles", "conv_traffic.json")
    t = json.load(open(tpath)) if os.path.exists(tpath) else {}
    t.update({"block32_FETCH_SIZE_KB_raw": f32["mean_KB"], "block32_WRITE_SIZE_KB": w32["mean_KB"],
              "block32_hbm_bytes_per_launch": 1024.0 * (2 * f32["mean_KB"] + w32["mean_KB"]),
              "block32_source": f"profiles/{ROUND}_final.sh -> collect_r04.py ({ROUND} kernels)"})
    json.dump(t, open(tpath, "w"), indent=1)
# rocprofv3's stats file aggregates by kernel NAME; the driver command's chrono_like_loop secondary also runs the pass in slices (pass
# schedule "one block ahead": same kernel, 224 workgroups instead of 256, a quarter of the bytes), so the name's row mixes both.  The
# kernel trace of the same run has the grid size of every dispatch: one row per (kernel, grid size) here.
def stats_by_grid(subdir, out_name):
    paths = sorted(glob.glob(os.path.join(SRC, subdir, "*", "*_kernel_trace.csv")), key=os.path.getmtime, reverse=True)
    paths = [q for q in paths if "hc::" in open(q).read()]
    if not paths:
        return {}
    acc = {}
    with open(paths[0]) as fh:
        for row in csv.DictReader(fh):
            if "hc::" not in row["Kernel_Name"]:
                continue
            key = (short(row["Kernel_Name"]), int(row["Grid_Size_X"]) // int(row["Workgroup_Size_X"]))
            acc.setdefault(key, []).append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    with open(os.path.join(DST, out_name), "w") as fh:
        fh.write('"Name","Workgroups","Calls","TotalDurationNs","AverageNs","MinNs","MaxNs"\n')
        for (name, wgs), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            fh.write(f'"{name}",{wgs},{len(v)},{sum(v)},{sum(v) / len(v):.1f},{min(v)},{max(v)}\n')
    return acc


by_grid = stats_by_grid("stats_default", "c3_driver_cmd_kernel_stats_by_grid.csv")
stats_by_grid("stats_c4rank", "c4_rank_share_kernel_stats_by_grid.csv")
# roofline.frac recomputed from the committed files alone: algorithmic bytes per launch (bench line) / AverageNs of the FULL pass
# (conv_block_kernel on 256 workgroups = one per CU)
try:
    bl = json.loads(open(os.path.join(DST, "bench_c3_driver_cmd_under_rocprof.json")).read())
    full = max((k for k in by_grid if k[0].startswith("hc::conv_block_kernel<6")), key=lambda k: k[1])
    v = by_grid[full]
    avg_ns = sum(v) / len(v)
    b = bl["roofline"]["algorithmic_bytes_per_launch"]
    rec = {"file": "c3_driver_cmd_kernel_stats_by_grid.csv", "kernel": full[0], "workgroups": full[1], "calls": len(v), "AverageNs": avg_ns,
           "algorithmic_bytes_per_launch": b, "frac_from_rocprofv3": b / (avg_ns * 1e-9) / 8e12,
           "frac_in_the_bench_line_of_that_run": bl["roofline"]["frac"],
           "frac_in_the_bench_line_without_the_tool": json.loads(open(os.path.join(DST, "bench_c3_driver_cmd.json")).read())["roofline"]["frac"],
           "note": "command: rocprofv3 --kernel-trace --stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-c4-share --no-c4-one-gpu; "
                   "under rocprofv3 the bench's own timings use HIP launches + HIP events, which include the gap in front of the kernel (DESIGN.md 3.4: "
                   "frac_in_the_bench_line_of_that_run); without the tool they are the completion-signal timestamps of the library's own AQL dispatches "
                   "(bench_c3_driver_cmd.json, same box, same script run: frac_in_the_bench_line_without_the_tool) -- the figure BENCH_rNN.json carries"}
    json.dump(rec, open(os.path.join(DST, "roofline_recomputed.json"), "w"), indent=1)
    print(json.dumps(rec, indent=1))
except Exception as e:  # noqa: BLE001
    print("roofline recomputation failed:", e)
print(json.dumps(pmc, indent=1)[:3000])
