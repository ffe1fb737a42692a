"""Turns the raw output of profiles/r02_final.sh (gpurun_out/r02final/) into the committed summaries under profiles/r02/ and
profiles/conv_traffic.json.  Usage: python profiles/collect_r02.py [gpurun_out/r02final]"""
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r02final")
DST = os.path.join(ROOT, "profiles", "r02")


def short(name):
    return re.sub(r"\(.*", "", name).replace("void ", "").strip()


def counter_means(subdir, counter):
    out = {}
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


def last_json(path):
    lines = [ln for ln in open(path) if ln.startswith("{")]
    return json.loads(lines[-1]) if lines else None


os.makedirs(DST, exist_ok=True)
shutil.copy(stats_file("stats_default"), os.path.join(DST, "c3_default_cmd_kernel_stats.csv"))
j = last_json(os.path.join(SRC, "stats_default.log"))
if j:
    json.dump(j, open(os.path.join(DST, "bench_c3_default_cmd_under_rocprof.json"), "w"))
for name in ("bench_c3_default.json", "bench_c3_depth16.json", "bench_c3_plain.json", "bench_c3_stepdt0.007.json", "bench_c4_1gpu.json",
             "host_path.json"):
    p = os.path.join(SRC, name)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(DST, name))

pmc = {k: counter_means(k, "FETCH_SIZE" if k.startswith("fetch") else "WRITE_SIZE") for k in ("fetch32", "write32", "fetch16", "write16", "fetch0", "write0")}
pmc["units"] = "KB as reported by rocprofv3 (raw); gfx950 correction for wide streaming reads: FETCH_SIZE x2"
json.dump(pmc, open(os.path.join(DST, "pmc_traffic.json"), "w"), indent=1)


def pick(d, prefix):
    return next((k for k in d if k.startswith(prefix)), None)


traffic = {
    "workload": "C3 (bench.py default)",
    "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for wide streaming reads -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact",
    "source": "profiles/r02_final.sh -> profiles/collect_r02.py -> profiles/r02/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes)",
}
k0 = pick(pmc["fetch0"], "hc::conv_step_kernel")
if k0:
    traffic.update({"kernel": k0 + " (plain step, --lookahead 0)", "FETCH_SIZE_KB_raw": pmc["fetch0"][k0]["mean_KB"],
                    "WRITE_SIZE_KB": pmc["write0"][k0]["mean_KB"],
                    "hbm_bytes_per_launch": 1024.0 * (2 * pmc["fetch0"][k0]["mean_KB"] + pmc["write0"][k0]["mean_KB"])})
for depth in (32, 16):
    kb = pick(pmc[f"fetch{depth}"], "hc::conv_block_kernel")
    if kb:
        traffic.update({f"block{depth}_kernel": kb + f" (look-ahead pass, one launch per {depth} steps)",
                        f"block{depth}_FETCH_SIZE_KB_raw": pmc[f"fetch{depth}"][kb]["mean_KB"],
                        f"block{depth}_WRITE_SIZE_KB": pmc[f"write{depth}"][kb]["mean_KB"],
                        f"block{depth}_hbm_bytes_per_launch": 1024.0 * (2 * pmc[f"fetch{depth}"][kb]["mean_KB"] + pmc[f"write{depth}"][kb]["mean_KB"])})
json.dump(traffic, open(os.path.join(ROOT, "profiles", "conv_traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
