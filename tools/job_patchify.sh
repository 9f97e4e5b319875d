#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/patchify; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_cabi.py -m gpu -q -x -k "patchify or cabi or uhd_forward or shipped" > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 600 python tools/bench_patchify.py > $O/bench_patchify.txt 2>&1; cat $O/bench_patchify.txt
for i in 1 2; do
  timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --timed-only 2> $O/bench.err | grep '^{' > $O/bench.json
  python - <<PY
import json
d=json.load(open("$O/bench.json")); print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
done
