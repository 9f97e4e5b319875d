#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3g; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -k "lfss or ss2d_core_golden or ss2d_core_backward or trainable" > $O/tests.log 2>&1; tail -15 $O/tests.log
python bench.py --no-cpu-baseline --no-train --no-bf16 --steps 10 > $O/bench_nocpu.json 2> $O/bench_nocpu.err
tail -3 $O/bench_nocpu.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r3g/bench_nocpu.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["ms_per_step"], (d["roofline"].get("isolated") or {}).get("ms_per_step"))
print({k:round(v["ms_per_step"],3) for k,v in d["roofline_table"].items()})
PY
