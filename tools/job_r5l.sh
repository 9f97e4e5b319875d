#!/bin/bash
O=gpurun_out/prof_round; mkdir -p $O
python -c "import wave_mamba_amd as wm; print('build_id', wm._lib.build_id())"
bash tools/profile_bench.sh $O/multi > $O/profile_multi.log 2>&1
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/prof_round/multi/bench_line.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])
PY
sed -n 2p $O/multi/bench_per_step_kernel_breakdown.txt | cut -c1-120
