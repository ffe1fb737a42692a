"""Aggregates rocprofv3 --pmc counter_collection CSVs (one directory per pass) into one JSON: mean counter value per hc:: kernel.
Usage: python profiles/collect_pmc.py out.json dir1 dir2 ..."""
import csv
import glob
import json
import os
import re
import sys

out, dirs = sys.argv[1], sys.argv[2:]
res = {}
for d in dirs:
    for path in sorted(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")), key=os.path.getmtime):
        acc = {}
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if "hc::" not in row["Kernel_Name"]:
                    continue
                k = re.sub(r"\(.*", "", row["Kernel_Name"]).strip()
                acc.setdefault(k, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        if not any("conv_" in k for k in acc):
            continue
        for k, cs in acc.items():
            for cname, vals in cs.items():
                res.setdefault(k, {})[cname] = sum(vals) / len(vals)
                res[k]["dispatches"] = len(vals)
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in res.items() if "conv_block" in k}, indent=1, sort_keys=True))
