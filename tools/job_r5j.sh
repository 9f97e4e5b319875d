#!/bin/bash
O=gpurun_out/r5j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lfss or golden or uhd" 2>&1 | tail -2 | tee $O/tests.txt
timeout 300 python tools/bench_lfss_rz.py 2>&1 | grep level | cut -c1-150 | tee $O/bench_lfss_rz.txt
for i in 1 2; do timeout 600 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --timed-only 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('images/s', round(d['value'], 3), 'ms', round(d['ms_per_step'], 3))"; done | tee $O/bench.txt
