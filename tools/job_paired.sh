#!/bin/bash
# paired-mode experiment: parity tests of the new mode + core regression subset, A/B per call, A/B of the UHD step
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/paired; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "paired or ss2d_core_vs_oracle or ss2d_core_bf16 or lfss_block" > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 600 python tools/bench_core_paired.py > $O/ab_per_call.txt 2>&1; cat $O/ab_per_call.txt
for p in 0 1 0 1; do
  WM_CORE_PAIRED=$p timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --timed-only 2> $O/bench_$p.err | grep '^{' > $O/bench_$p.json
  python - <<PY
import json
d=json.load(open("$O/bench_$p.json")); print("paired=$p", d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
done
