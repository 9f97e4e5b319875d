#!/bin/bash
# round evidence on the current build: profile_round + the default bench line
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/final; mkdir -p $O
timeout 2400 bash tools/profile_round.sh > $O/profile_round.log 2>&1; tail -30 $O/profile_round.log | head -14
timeout 1500 python bench.py 2> $O/bench_default.err | grep '^{' > $O/bench_default.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/final/bench_default.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["ms_per_step"], (d["roofline"].get("isolated") or {}).get("ms_per_step"), d["roofline"]["traffic"], d["roofline"].get("traffic_over_algorithmic"))
print({k:round(v["ms_per_step"],3) for k,v in d["roofline_table"].items()})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "parity", d["parity"]["rel_l2_vs_cpu_oracle"], d["parity"]["abs_dpsnr_db"])
t=d["training_config3_one_gpu"]; print("train", t["images_per_s"], t["ms_per_step"], t["selective_scan_backward"]["ms_per_step"], t["selective_scan_backward"]["frac"], t["first_step_loss_parity"]["rel_diff"])
print("concurrent", d["concurrent_forwards"]["images_per_s"], "hot_path_sum", d.get("hot_path_sum"))
PY
