for rows in 12 6; do for un in 1 2; do for wgs in 2048 4096 8192; do
  echo "rows=$rows unroll=$un target_wgs=$wgs"
  HC_CONV_ROWS=$rows HC_CONV_UNROLL=$un HC_CONV_TARGET_WGS=$wgs python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms/step %.4f  conv_us %.1f  GB/s %.0f'%(d['ms_per_step'], d['roofline']['mean_kernel_us'], d['roofline']['achieved']))"
done; done; done
