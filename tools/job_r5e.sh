#!/bin/bash
O=gpurun_out/final; mkdir -p $O
timeout 1500 python bench.py 2> $O/bench_default.err | grep '^{' > $O/bench_default.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/final/bench_default.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"].get("traffic_over_algorithmic"))
print({k:(round(v["ms_per_step"],3), round(v.get("frac",0),3)) for k,v in d["roofline_table"].items()})
t=d["training_config3_one_gpu"]; print("train", t["images_per_s"], t["ms_per_step"])
PY
