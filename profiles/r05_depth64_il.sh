#!/bin/bash
# round 5: the second (interleaved-issue) form of the depth-64 pass, HC_BLOCK64_R=11 / 12 (3 / 4 register slots): parity of the two
# depth-64 tests with it, then the kernel alone beside the first form (R = 3) at C3 and at one C4/8 rank, then its SQ counters.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
for r in 11 12; do
  for mt in 6 4; do
    echo "== parity: HC_BLOCK64_MT=$mt HC_BLOCK64_R=$r"
    HC_BLOCK64_MT=$mt HC_BLOCK64_R=$r timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "depth_64 or depth64" 2>&1 | tail -2
  done
done > $O/depth64_il_parity.txt 2>&1
cat $O/depth64_il_parity.txt
for r in 3 11 12; do
  for mt in 6 4; do
    for pause in 500 0; do
      echo "== HC_BLOCK64_MT=$mt HC_BLOCK64_R=$r HC_TUNING_PASS_PAUSE_US=$pause"
      HC_BLOCK64_MT=$mt HC_BLOCK64_R=$r HC_TUNING_PASS_PAUSE_US=$pause python profiles/pass_depth_probe.py 2>/dev/null | grep "depth 64"
    done
  done
done > $O/depth64_il_probe.txt 2>&1
cat $O/depth64_il_probe.txt
cd /tmp && export TMPDIR=/tmp
export HC_BLOCK64_MT=6 HC_BLOCK64_R=11
L=64
B="python3 $R/bench.py --steps 192 --warmup 8 --no-cpu-baseline --no-secondary --profile-stride 1000000 --lookahead $L"
rm -rf /tmp/pmc_il_*
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_il_1 -- $B > /tmp/pmc_il_1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_il_3 -- $B > /tmp/pmc_il_3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmc_il_5 -- $B > /tmp/pmc_il_5.log 2>&1
python3 $R/profiles/collect_pmc.py $O/pass_pmc_depth64_il.json /tmp/pmc_il_1 /tmp/pmc_il_3 > /dev/null
cp $(ls /tmp/pmc_il_5/*/*kernel_stats.csv | head -1) $O/pass_depth64_il_kernel_stats.csv
grep conv_block $O/pass_depth64_il_kernel_stats.csv
python3 - <<'PY'
import json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r05")
d = json.load(open(f"{O}/pass_pmc_depth64_il.json"))
for k in d:
    if "conv_block" in k:
        print(k)
        for n in sorted(d[k]):
            print(f"   {n:32s} {d[k][n]:16.1f}")
PY
