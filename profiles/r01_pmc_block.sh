R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/counters_list.txt 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_block1 -- python3 $R/bench.py --steps 48 --warmup 4 --no-cpu-baseline --profile-stride 1000000 > $R/gpurun_out/pmc_block1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_block2 -- python3 $R/bench.py --steps 48 --warmup 4 --no-cpu-baseline --profile-stride 1000000 > $R/gpurun_out/pmc_block2.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_block3 -- python3 $R/bench.py --steps 48 --warmup 4 --no-cpu-baseline --profile-stride 1000000 > $R/gpurun_out/pmc_block3.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmc_block4 -- python3 $R/bench.py --steps 48 --warmup 4 --no-cpu-baseline --profile-stride 1000000 > $R/gpurun_out/pmc_block4.log 2>&1
