import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print({k:d.get(k) for k in ("value","ms_per_step","median_ms_per_step","scaling","workload")})
print("steady", d.get("steady_state"))
print("pipe", d.get("device_pipelined"))
print("c4_one_gpu", json.dumps(d.get("c4_one_gpu")))
c=d.get("c4_rank_share") or {}
print("c4_rank_share", c.get("ms_per_step"), c.get("per_step_us"), c.get("back_to_back_pass_at_block_start"))
print("roofline", d["roofline"]["frac"], d["roofline"]["mean_kernel_us"], d["roofline"].get("profile_file"))
print("parity", d.get("parity_max_rel_err_vs_oracle"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
