#!/bin/bash
O=gpurun_out/final; mkdir -p $O
timeout 1500 python bench.py 2> $O/bench_default.err | grep '^{' > $O/bench_default.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/final/bench_default.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"].get("traffic_over_algorithmic"), (d["roofline"].get("isolated") or {}).get("frac"))
t=d["training_config3_one_gpu"]; print("train", t["images_per_s"], t["ms_per_step"]); print("hot", d["hot_path_sum"]["frac"], "concurrent", d["concurrent_forwards"]["images_per_s"], "bf16", d["bf16_storage"]["images_per_s"], "cpu", d["cpu_baseline"]["value"])
PY
