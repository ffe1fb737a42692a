#!/bin/bash
# rocprofv3 kernel trace + stats of the default bench command (round 2).  Usage: bash profiles/r02_trace.sh <tag>
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 2000 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/${tag}_rocprof.err
find /tmp/prof_$tag -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.csv \;
find /tmp/prof_$tag -name "*kernel_trace.csv" -exec cp {} /tmp/${tag}_kernel_trace.csv \;
cp /tmp/${tag}_kernel_trace.csv $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_trace.csv
python3 - <<PY
import csv
rows = list(csv.DictReader(open("/tmp/${tag}_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# timeline of the synchronous timed region: take 48 consecutive dispatches from the middle
mid = len(rows) // 3
t0 = int(rows[mid]["Start_Timestamp"])
out = []
prev_end = None
for r in rows[mid:mid + 60]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    out.append("%9.2f us  dur %8.2f us  gap %7.2f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, r["Kernel_Name"][:60]))
    prev_end = e
open("$GRAFT_REPO_ROOT/gpurun_out/${tag}_timeline.txt", "w").write("\n".join(out) + "\n")
PY
head -12 $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.csv | cut -c1-150
