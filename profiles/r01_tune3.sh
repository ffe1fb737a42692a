run() { python bench.py --no-cpu-baseline --steps 320 --warmup 32 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   evals/s %d  ms/step %.5f  block_us %.1f  step_us %.2f'%(d['value'], d['ms_per_step'], d['roofline']['mean_kernel_us'], d['roofline']['in_block_step_kernel_us']))"; }
echo "default"; run
echo "HC_REM_NT=1"; HC_REM_NT=1 run
for r in 8 32 64; do echo "HC_REM_CHUNK_GP=$r"; HC_REM_CHUNK_GP=$r run; done
for e in 8 32; do echo "HC_EXC_CHUNK_GP=$e"; HC_EXC_CHUNK_GP=$e run; done
echo "HC_CONV_UNROLL=1"; HC_CONV_UNROLL=1 run
echo "HC_CONV_UNROLL=3"; HC_CONV_UNROLL=3 run
