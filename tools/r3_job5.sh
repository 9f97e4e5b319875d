#!/bin/bash
# round-3 GPU job 5: micro-trimmed core (rcp-free softplus, scalar tile mask), recalibrated planner: parity, timing, timeline
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3e; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -k "ss2d_core or lfss_block or core_abi or scan_golden or scan_vs_oracle or scan_stress" > $O/tests.log 2>&1; tail -5 $O/tests.log
python tools/core_accuracy.py > $O/acc.log 2>&1
python tools/bench_core.py --iters 5 > $O/core_new.log 2>&1
WAVEMAMBA_HIP_AB=1 WAVEMAMBA_HIP_LIB=build/variants/r2.so python tools/bench_core.py --iters 5 > $O/core_r2.log 2>&1
python tools/bench_core.py --iters 3 --dstate 32 --levels 1 > $O/core_new_n32.log 2>&1
for l in 1 2 3; do WAVEMAMBA_HIP_LIB=build/variants/stamp2.so python tools/core_stamps.py --level $l > $O/stamps_l$l.log 2>&1; done
python bench.py --no-cpu-baseline --steps 10 > $O/bench_nocpu.json 2> $O/bench_nocpu.err
cat $O/acc.log $O/core_*.log $O/stamps_*.log; python - <<'PY'
import json
d=json.load(open("gpurun_out/r3e/bench_nocpu.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["ms_per_step"], (d["roofline"].get("isolated") or {}).get("ms_per_step"))
print({k:round(v["ms_per_step"],3) for k,v in d["roofline_table"].items()}); print(d.get("bf16_storage"))
PY
