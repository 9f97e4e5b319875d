#!/bin/bash
# fewer core waves per compute unit -> room for the side streams' kernels?  A/B of the whole step
O=gpurun_out/r4t; mkdir -p $O
for v in "" build/variants/nw12.so build/variants/nw8x1.so; do
  export WAVEMAMBA_HIP_LIB=$v; [ -z "$v" ] && unset WAVEMAMBA_HIP_LIB
  echo "=== lib: ${v:-shipped}" | tee -a $O/ab.txt
  timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ss2d_core and not backward and not bwd" 2>&1 | tail -3 | tee -a $O/ab.txt
  timeout 300 python tools/bench_core.py --iters 10 2>&1 | grep level | tee -a $O/ab.txt
  for ts in 1 0; do
    echo "-- WM_TWO_STREAMS=$ts" | tee -a $O/ab.txt
    WM_TWO_STREAMS=$ts timeout 600 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --timed-only 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('images/s', d['value'], 'ms', d['ms_per_step'], 'roofline', d.get('roofline', {}).get('frac'))" | tee -a $O/ab.txt
  done
done
