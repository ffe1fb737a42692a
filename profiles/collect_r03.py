"""Turns the raw output of profiles/r03_final.sh (gpurun_out/r03final/) into the committed summaries under profiles/r03/.
Usage: python profiles/collect_r03.py [gpurun_out/r03final]"""
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r03final")
DST = os.path.join(ROOT, "profiles", "r03")
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
              "block32_source": "profiles/r03_final.sh -> collect_r03.py (round 3 kernels)"})
    json.dump(t, open(tpath, "w"), indent=1)
print(json.dumps(pmc, indent=1)[:6000])
