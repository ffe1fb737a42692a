#!/bin/bash
# round 5: where the depth-64 pass loses its matrix-pipe time -- the interleaved form (HC_BLOCK64_R=11) beside three timing bounds of it
# (13: the loop issues no gathers, 14: no loads at all, 15: MFMAs only; results wrong), kernel alone at C3 and at one C4/8 rank, then
# the SQ counters of form 11 and of bound 15.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
echo "== parity of the interleaved form (HC_BLOCK64_MT=6 HC_BLOCK64_R=11), first failure in full"
HC_BLOCK64_MT=6 HC_BLOCK64_R=11 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "depth_64 or depth64" 2>&1 | tail -30 > $O/depth64_il_parity_detail.txt
tail -30 $O/depth64_il_parity_detail.txt
for r in 11 13 14 15; do
  echo "== HC_BLOCK64_MT=6 HC_BLOCK64_R=$r HC_TUNING_PASS_PAUSE_US=500"
  HC_BLOCK64_MT=6 HC_BLOCK64_R=$r HC_TUNING_PASS_PAUSE_US=500 python profiles/pass_depth_probe.py 2>/dev/null | grep "depth 64"
done > $O/depth64_bisect.txt 2>&1
cat $O/depth64_bisect.txt
cd /tmp && export TMPDIR=/tmp
export HYDROCHRONO_AMD_FLAVOR=tuning HC_BLOCK64_MT=6 HC_TUNING_PASS_PAUSE_US=500
for r in 11 15; do
  export HC_BLOCK64_R=$r
  rm -rf /tmp/pmc_b${r}_*
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_b${r}_1 -- python3 $R/profiles/pass_depth_probe.py > /tmp/pmc_b${r}_1.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d /tmp/pmc_b${r}_2 -- python3 $R/profiles/pass_depth_probe.py > /tmp/pmc_b${r}_2.log 2>&1
  python3 - $r <<'PY'
import csv, glob, sys, os, json
r = sys.argv[1]
res = {}
for d in (f"/tmp/pmc_b{r}_1", f"/tmp/pmc_b{r}_2"):
    for path in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"]
            if "conv_block_kernel<6, %s, 4" % r not in k:
                continue
            key = (k.split("(")[0], int(row["Grid_Size"]) if "Grid_Size" in row else 0)
            res.setdefault(key, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
out = {}
for (k, g), cs in res.items():
    out[f"{k} grid={g}"] = {n: sum(v) / len(v) for n, v in cs.items()}
    out[f"{k} grid={g}"]["dispatches"] = max(len(v) for v in cs.values())
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r05")
json.dump(out, open(f"{O}/pass_pmc_depth64_form{r}.json", "w"), indent=1, sort_keys=True)
for k in out:
    print(k)
    for n in sorted(out[k]):
        print(f"   {n:32s} {out[k][n]:16.1f}")
PY
done
