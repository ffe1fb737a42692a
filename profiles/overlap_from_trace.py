"""Reads a rocprofv3 kernel trace (csv) and reports how much of the look-ahead passes' run time overlaps step kernels of another queue.
python profiles/overlap_from_trace.py <dir with *_kernel_trace.csv>"""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")))
rows.sort(key=lambda r: r[1])
passes = [r for r in rows if "conv_block_kernel<6" in r[0]]
steps = [r for r in rows if "finalize_kernel" in r[0]]
queues = {}
for r in rows:
    queues.setdefault(r[3], {}).setdefault(r[0].split("(")[0][:60], 0)
    queues[r[3]][r[0].split("(")[0][:60]] += 1
print("kernels per queue:")
for q, d in queues.items():
    print("  queue", q, {k: v for k, v in sorted(d.items(), key=lambda kv: -kv[1])[:6]})
inside = 0
for s in steps:
    if any(p[1] <= s[1] and s[2] <= p[2] and p[3] != s[3] for p in passes):
        inside += 1
tot_pass = sum(p[2] - p[1] for p in passes) / 1e3
print(f"{len(passes)} launches of the pass kernel, {tot_pass:.0f} us in total (mean {tot_pass / max(1, len(passes)):.1f} us); "
      f"{len(steps)} step kernels, {inside} of them ran entirely INSIDE a pass launch of another queue")
if passes:
    p = passes[len(passes) // 2]
    print("one pass launch and the kernels that started while it ran:")
    print(f"  [{0:8.1f} .. {(p[2] - p[1]) / 1e3:8.1f} us] queue {p[3]} {p[0][:50]}")
    for r in rows:
        if p[1] < r[1] < p[2] and r is not p:
            print(f"  [{(r[1] - p[1]) / 1e3:8.1f} .. {(r[2] - p[1]) / 1e3:8.1f} us] queue {r[3]} {r[0][:50]}")
