#!/bin/bash
# A/B of the look-ahead pass kernels and a kernel trace of the synchronous bench (round 2).  Usage: bash profiles/r02_ab.sh
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("evals/s %.0f  mean %.2f us  median %.2f us  pass %.1f us frac %.3f  step %.2f us scatter %.2f us" % (d["value"], d["ms_per_step"]*1e3, d["median_ms_per_step"]*1e3, r["mean_kernel_us"], r["frac"], r["step_kernel_us"], r["scatter_kernel_us"]))'
run() { python bench.py --no-cpu-baseline --no-secondary --steps 1600 2>/dev/null | tail -1 | python -c "$P"; }
for rep in 1 2; do
  echo -n "v1 (round-1 kernel):      "; HC_BLOCK_KERNEL=1 run
  for d in 2 3 4; do echo -n "v2 rolling, depth $d:      "; HC_BLOCK_DEPTH=$d run; done
done
