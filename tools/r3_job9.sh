#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3i; mkdir -p $O
python -m pytest tests -m gpu -q --durations=12 > $O/tests_full.log 2>&1; tail -30 $O/tests_full.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3i/bench_default.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["ms_per_step"], (d["roofline"].get("isolated") or {}).get("ms_per_step"))
print({k:round(v["ms_per_step"],3) for k,v in d["roofline_table"].items()})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "parity", d["parity"])
print("bf16", {k:v for k,v in d["bf16_storage"].items() if k!="note"})
t=d["training_config3_one_gpu"]; print("train", {k:t[k] for k in t if k not in ("workload",)})
print("concurrent", d["concurrent_forwards"]["images_per_s"])
PY
