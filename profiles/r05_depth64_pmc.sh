#!/bin/bash
# round 5: SQ / GRBM counters of the depth-64 pass (conv_block_kernel<6, 3, 4, 1>) beside the shipped depth-32 pass (<6, 4, 2, 1>) at C3:
# what the matrix pipe, the vector issue and the waits take, and the clock the chip holds (GRBM_GUI_ACTIVE / 8 / kernel time).
# Separate --pmc passes with --kernel-trace only (MI355X_MICROARCH.md).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export HC_BLOCK64_MT=6 HC_BLOCK64_R=3
for L in 32 64; do
  B="python3 $R/bench.py --steps 192 --warmup 8 --no-cpu-baseline --no-secondary --profile-stride 1000000 --lookahead $L"
  rm -rf /tmp/pmc_d${L}_*
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_d${L}_1 -- $B > /tmp/pmc_d${L}_1.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d /tmp/pmc_d${L}_2 -- $B > /tmp/pmc_d${L}_2.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_d${L}_3 -- $B > /tmp/pmc_d${L}_3.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_d${L}_4 -- $B > /tmp/pmc_d${L}_4.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmc_d${L}_5 -- $B > /tmp/pmc_d${L}_5.log 2>&1
  python3 $R/profiles/collect_pmc.py $O/pass_pmc_depth${L}.json /tmp/pmc_d${L}_1 /tmp/pmc_d${L}_2 /tmp/pmc_d${L}_3 /tmp/pmc_d${L}_4 > /dev/null
  cp $(ls /tmp/pmc_d${L}_5/*/*kernel_stats.csv | head -1) $O/pass_depth${L}_kernel_stats.csv
  tail -2 /tmp/pmc_d${L}_1.log
done
python3 - <<'PY'
import json, os, csv
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r05")
for L in (32, 64):
    d = json.load(open(f"{O}/pass_pmc_depth{L}.json"))
    k = [x for x in d if "conv_block" in x][0]
    c = d[k]
    us = None
    for row in csv.DictReader(open(f"{O}/pass_depth{L}_kernel_stats.csv")):
        if "conv_block" in row["Name"]:
            us = float(row["AverageNs"]) / 1e3
    print(k, "avg us", us)
    for n in sorted(c):
        print(f"   {n:32s} {c[n]:16.1f}")
    if us:
        clk = c.get("GRBM_GUI_ACTIVE", 0) / 8 / us / 1e3
        print(f"   effective clock (GRBM_GUI_ACTIVE / 8 / kernel time): {clk:.2f} GHz")
        print(f"   MFMA busy cycles per SIMD / kernel cycles at that clock: {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / (us * 1e3 * clk):.3f}")
PY
