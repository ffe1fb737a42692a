#!/bin/bash
# r06_tail.sh -- (a) is the slow tenth of the steps in the --step-dt 0.007 line (p90 31 us in the last full run, 12.7 us in the run before)
# a property of the box or of the bench's raw hc_step_many entry / thread binding?  (b) where the GPU suite's eight minutes go.
O=gpurun_out/r06tail; mkdir -p $O
for v in "" "--no-pin"; do
  for i in 1 2 3; do
    echo "== bench.py --step-dt 0.007 $v (run $i)" >> $O/stepdt_tail.txt
    timeout 300 python bench.py --step-dt 0.007 --no-cpu-baseline --no-secondary --no-c4-share --no-c4-one-gpu --no-small-configs --no-init $v 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('value', round(j['value']), 'mean', round(j['ms_per_step']*1e3, 2), 'median', round(j['median_ms_per_step']*1e3, 2), 'p10', round(j['p10_ms_per_step']*1e3, 2), 'p90', round(j['p90_ms_per_step']*1e3, 2), 'passes', j['passes_in_timed_region'], 'aql', j['aql_dispatches'])
" >> $O/stepdt_tail.txt
  done
done
for i in 1 2; do
  echo "== bench.py (dt 0.01) run $i" >> $O/stepdt_tail.txt
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-c4-share --no-c4-one-gpu --no-small-configs --no-init 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('value', round(j['value']), 'mean', round(j['ms_per_step']*1e3, 2), 'median', round(j['median_ms_per_step']*1e3, 2), 'p10', round(j['p10_ms_per_step']*1e3, 2), 'p90', round(j['p90_ms_per_step']*1e3, 2), 'passes', j['passes_in_timed_region'], 'aql', j['aql_dispatches'])
" >> $O/stepdt_tail.txt
done
cat $O/stepdt_tail.txt
timeout 1200 python -m pytest tests -m gpu -x -q --durations=40 > $O/pytest_durations.txt 2>&1
tail -60 $O/pytest_durations.txt
