cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
python bench.py --gpus 2 --steps 64 --warmup 33 > gpurun_out/r05/bench_c4_single_process_2ctx_one_gpu.json 2> gpurun_out/r05/sp2.err; echo "single-process rc=$?"; tail -2 gpurun_out/r05/sp2.err
for ex in host rccl; do
HC_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29621 bench.py --gpus 2 --steps 40 --warmup 8 --exchange $ex > gpurun_out/r05/bench_c4_2ranks_share_gpu_$ex.json 2> gpurun_out/r05/r2_$ex.err; echo "2 ranks ($ex) rc=$?"; tail -3 gpurun_out/r05/r2_$ex.err
done
python - <<'PY'
import json
for f in ("bench_c4_single_process_2ctx_one_gpu","bench_c4_2ranks_share_gpu_host","bench_c4_2ranks_share_gpu_rccl"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/r05/{f}.json") if l.startswith("{")][-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, d["value"], d["ms_per_step"], d.get("exchange_check"), d.get("per_rank_ms_per_step"), (d.get("other_exchange_mode") or {}).get("collective_world_size"), (d.get("other_exchange_mode") or {}).get("error"))
    for p in d.get("per_rank", []): print("   ", {k: (round(v,3) if isinstance(v,float) else v) for k,v in p.items() if k != "kernel_us_per_step"}, {k: round(v,1) for k,v in p["kernel_us_per_step"].items() if k!="note"})
PY
