"""Turns the raw output of profiles/r01_final.sh (gpurun_out/final/) into the committed summaries under profiles/r01/ and
profiles/conv_traffic.json.  Usage: python profiles/collect_final.py [gpurun_out/final]"""
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "final")
DST = os.path.join(ROOT, "profiles", "r01")


def short(name):
    return re.sub(r"\(.*", "", name).strip()


def counter_means(subdir, counter):
    """mean counter value per kernel (hc:: kernels only) over all dispatches of the profiled process"""
    out = {}
    # gpurun merges every call's output into gpurun_out/: take the newest run's file that holds the hc:: kernels
    for path in sorted(glob.glob(os.path.join(SRC, subdir, "*", "*_counter_collection.csv")), key=os.path.getmtime):
        acc = {}
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter or "hc::" not in row["Kernel_Name"]:
                    continue
                acc.setdefault(short(row["Kernel_Name"]), []).append(float(row["Counter_Value"]))
        if any("conv_" in k for k in acc):
            out = {k: {"dispatches": len(v), "mean_KB": sum(v) / len(v)} for k, v in acc.items()}
    return out


def stats_file(subdir):
    for path in sorted(glob.glob(os.path.join(SRC, subdir, "*", "*_kernel_stats.csv")), key=os.path.getmtime, reverse=True):
        if "hc::conv_" in open(path).read():
            return path
    raise SystemExit(f"no kernel stats with hc:: kernels under {subdir}")


os.makedirs(DST, exist_ok=True)
shutil.copy(stats_file("stats"), os.path.join(DST, "final_c3_lookahead_kernel_stats.csv"))
shutil.copy(stats_file("stats_plain"), os.path.join(DST, "final_c3_plain_kernel_stats.csv"))
if os.path.isdir(os.path.join(SRC, "stats_default")):  # rocprofv3 around the default command line, `python3 bench.py`
    shutil.copy(stats_file("stats_default"), os.path.join(DST, "final_c3_default_cmd_kernel_stats.csv"))
    with open(os.path.join(SRC, "stats_default.log")) as fh:
        lines = [ln for ln in fh if ln.startswith("{")]
    if lines:
        open(os.path.join(DST, "final_bench_c3_default_cmd_under_rocprof.json"), "w").write(lines[-1])
for src, dst in (("bench_c3.json", "final_bench_c3_lookahead.json"), ("bench_c3_plain.json", "final_bench_c3_plain.json"),
                 ("bench_c4_1gpu.json", "final_bench_c4_1gpu_lookahead.json"), ("host_path.json", "host_path.json")):
    shutil.copy(os.path.join(SRC, src), os.path.join(DST, dst))

pmc = {"fetch": counter_means("fetch", "FETCH_SIZE"), "write": counter_means("write", "WRITE_SIZE"),
       "fetch_plain": counter_means("fetch_plain", "FETCH_SIZE"), "write_plain": counter_means("write_plain", "WRITE_SIZE"),
       "units": "KB as reported by rocprofv3 (raw); gfx950 correction for wide streaming reads: FETCH_SIZE x2"}
json.dump(pmc, open(os.path.join(DST, "final_pmc.json"), "w"), indent=1)

stp = "void hc::conv_step_kernel<4, 2, true>"
blk = next(k for k in pmc["fetch"] if k.startswith("void hc::conv_block_kernel<"))  # <6> at C3 (6 row tiles per workgroup)
bench = json.loads(open(os.path.join(SRC, "bench_c3.json")).read())
plain = json.loads(open(os.path.join(SRC, "bench_c3_plain.json")).read())
traffic = {
    "workload": "C3 (bench.py default)",
    "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for wide streaming reads -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact",
    "kernel": "hc::conv_step_kernel<4,2,true> (plain step, --lookahead 0)",
    "FETCH_SIZE_KB_raw": pmc["fetch_plain"][stp]["mean_KB"],
    "WRITE_SIZE_KB": pmc["write_plain"][stp]["mean_KB"],
    "hbm_bytes_per_launch": 1024.0 * (2 * pmc["fetch_plain"][stp]["mean_KB"] + pmc["write_plain"][stp]["mean_KB"]),
    "algorithmic_bytes_per_launch": plain["roofline"]["algorithmic_bytes_per_launch"],
    "block_kernel": blk.replace("void ", "") + " (look-ahead pass, one launch per 16 steps)",
    "block_FETCH_SIZE_KB_raw": pmc["fetch"][blk]["mean_KB"],
    "block_WRITE_SIZE_KB": pmc["write"][blk]["mean_KB"],
    "block_hbm_bytes_per_launch": 1024.0 * (2 * pmc["fetch"][blk]["mean_KB"] + pmc["write"][blk]["mean_KB"]),
    "block_algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
    "source": "profiles/r01_final.sh -> profiles/collect_final.py -> profiles/r01/final_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes)",
}
json.dump(traffic, open(os.path.join(ROOT, "profiles", "conv_traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
