#!/bin/bash
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("evals/s %.0f  mean %.2f us  median %.2f us  pass %.1f us frac %.3f  step %.2f us scatter %.2f us" % (d["value"], d["ms_per_step"]*1e3, d["median_ms_per_step"]*1e3, r["mean_kernel_us"], r["frac"], r["step_kernel_us"], r["scatter_kernel_us"]))'
run() { python bench.py --no-cpu-baseline --no-secondary --steps 1600 "$@" 2>/dev/null | tail -1 | python -c "$P"; }
for rep in 1 2; do
  for r in 4 5 6 7; do echo -n "depth 32, R=$r: "; HC_BLOCK_R32=$r run --lookahead 32; done
done
