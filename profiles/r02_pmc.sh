#!/bin/bash
# SQ / TCC counters of the look-ahead pass (separate passes, rocprofv3 --pmc + --kernel-trace only).  Usage: bash profiles/r02_pmc.sh <tag> <bench args...>
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 96 --warmup 8 --no-cpu-baseline --no-secondary --profile-stride 1000000 $@"
rm -rf /tmp/pmc_${tag}_*
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_${tag}_1 -- $B > /tmp/pmc_${tag}_1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d /tmp/pmc_${tag}_2 -- $B > /tmp/pmc_${tag}_2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_${tag}_3 -- $B > /tmp/pmc_${tag}_3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_${tag}_4 -- $B > /tmp/pmc_${tag}_4.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace --output-format csv -d /tmp/pmc_${tag}_5 -- $B > /tmp/pmc_${tag}_5.log 2>&1
python3 $R/profiles/collect_pmc.py $R/gpurun_out/pmc_${tag}.json /tmp/pmc_${tag}_1 /tmp/pmc_${tag}_2 /tmp/pmc_${tag}_3 /tmp/pmc_${tag}_4 /tmp/pmc_${tag}_5
