#!/bin/bash
O=gpurun_out/r5f; mkdir -p $O; rm -f $O/bench.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv2d_ln or hfe or golden or uhd or model" 2>&1 | tail -4 | tee $O/tests.txt
for f in 1 0 1 0; do
  echo "== WM_FUSE_LN_CONV=$f" | tee -a $O/bench.txt
  WM_FUSE_LN_CONV=$f timeout 600 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-train --no-bf16 --concurrent 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('images/s', round(d['value'], 3), 'ms', round(d['ms_per_step'], 3), {k: round(v['ms_per_step'], 3) for k, v in d['roofline_table'].items() if k in ('conv1x1', 'layernorm2d')})" | tee -a $O/bench.txt
done
