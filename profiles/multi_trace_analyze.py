"""Reads a rocprofv3 --kernel-trace csv of profiles/multi_path_c and prints, for the runs with G contexts, how the step kernels of the
contexts of one hc_step_multi lie in time: start offsets between queues, durations, and the gap between consecutive steps per queue."""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        if "finalize_kernel" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"])))
rows.sort()
# split into runs by large gaps (context creation between the G = 1, 2, 4, 8 runs)
runs, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if b[0] - a[1] > 50_000_000:
        runs.append(cur)
        cur = []
    cur.append(b)
runs.append(cur)
for run in runs:
    queues = sorted({q for _, _, q in run})
    tail = run[len(run) // 2:]
    dur = defaultdict(list)
    for s, e, q in tail:
        dur[q].append((e - s) / 1e3)
    # group kernels into calls: kernels whose starts lie within 20 us of the first of the group
    groups, g = [], [tail[0]]
    for k in tail[1:]:
        if k[0] - g[0][0] < 20_000 and len(g) < len(queues):
            g.append(k)
        else:
            groups.append(g)
            g = [k]
    full = [g for g in groups if len(g) == len(queues)]
    if not full:
        continue
    spread = sorted((max(k[0] for k in g) - min(k[0] for k in g)) / 1e3 for g in full)
    span = sorted((max(k[1] for k in g) - min(k[0] for k in g)) / 1e3 for g in full)
    period = sorted((b[0][0] - a[0][0]) / 1e3 for a, b in zip(full, full[1:]))
    med = lambda v: v[len(v) // 2]  # noqa: E731
    print(f"G = {len(queues)}: step kernel duration per queue (median us): " + " ".join(f"{med(sorted(v)):.1f}" for v in dur.values()) +
          f" | start spread across queues {med(spread):.1f} us | first start -> last end {med(span):.1f} us | period between calls {med(period):.1f} us")
