#!/bin/bash
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train --no-bf16 --concurrent 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('images/s', round(d['value'], 3)); t = d['roofline_table']; print(json.dumps({k: t[k] for k in ('conv3x3', 'conv1x1')}, indent=1))"
