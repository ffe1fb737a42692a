#!/bin/bash
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("evals/s %.0f  mean %.2f us  median %.2f us  pass %.1f us frac %.3f  %s" % (d["value"], d["ms_per_step"]*1e3, d["median_ms_per_step"]*1e3, r["mean_kernel_us"], r["frac"], d.get("parity_max_rel_err_vs_oracle")))'
run() { python bench.py --no-cpu-baseline --no-secondary --steps 1600 "$@" 2>/dev/null | tail -1 | python -c "$P"; }
for rep in 1 2; do
  echo -n "d32 1 wave/SIMD R4:   "; run --lookahead 32
  echo -n "d32 2 waves/SIMD R2:  "; HC_BLOCK_V32=6 run --lookahead 32
  echo -n "d32 2 waves/SIMD R3:  "; HC_BLOCK_V32=7 run --lookahead 32
done
