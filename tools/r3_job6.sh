#!/bin/bash
# round-3 GPU job 6: everything new since job 4 - trimmed core, folded depth-wise conv, N = 32 backward, fp32 training
# convolutions, optimizer / checkpoint tests - then timing, timeline, NW = 8 experiments, bench
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3f; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -k "ss2d_core or lfss or core_abi or scan_backward or training_step or trainable or optimizer or two_training or checkpoint or conv2d_train" > $O/tests_new.log 2>&1; tail -25 $O/tests_new.log
python tools/core_accuracy.py > $O/acc.log 2>&1
python tools/bench_core.py --iters 5 > $O/core_new.log 2>&1
for v in nw8 nw8i; do WAVEMAMBA_HIP_LIB=build/variants/$v.so python tools/bench_core.py --iters 5 > $O/core_$v.log 2>&1; done
python tools/bench_core.py --iters 3 --dstate 32 --levels 1 > $O/core_new_n32.log 2>&1
for l in 1 2 3; do WAVEMAMBA_HIP_LIB=build/variants/stamp2.so python tools/core_stamps.py --level $l > $O/stamps_l$l.log 2>&1; done
python tools/grad_deviation.py > $O/grad_dev.log 2>&1
python bench.py --no-cpu-baseline --steps 10 > $O/bench_nocpu.json 2> $O/bench_nocpu.err
cat $O/acc.log $O/core_*.log $O/stamps_*.log; tail -8 $O/grad_dev.log; tail -3 $O/bench_nocpu.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r3f/bench_nocpu.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["ms_per_step"], (d["roofline"].get("isolated") or {}).get("ms_per_step"))
print({k:round(v["ms_per_step"],3) for k,v in d["roofline_table"].items()}); print(d.get("bf16_storage")); print(d.get("training_config3_one_gpu"))
PY
