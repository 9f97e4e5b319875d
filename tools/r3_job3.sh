#!/bin/bash
# round-3 GPU job 3: full GPU suite with the float64-truth criteria, gradient errors with the fp32 projection, bench
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3c; mkdir -p $O
python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_parity.py::test_training_step_per_parameter_gradients_on_gpu 2>&1 | tail -25 > $O/tests.log
python -m pytest tests -m gpu -q -k "training_step or backward" 2>&1 | tail -30 > $O/tests_grad.log
WAVEMAMBA_HIP_LIB=build/variants/f32proj.so python tools/grad_deviation.py > $O/grad_dev_f32proj.log 2>&1
WM_NO_HIP_CONV=1 WAVEMAMBA_HIP_LIB=build/variants/f32proj.so python tools/grad_deviation.py > $O/grad_dev_f32proj_noconv.log 2>&1
python bench.py --no-cpu-baseline --steps 10 > $O/bench_nocpu.json 2> $O/bench_nocpu.err
cat $O/tests.log $O/tests_grad.log $O/grad_dev_f32proj.log $O/grad_dev_f32proj_noconv.log; python - <<'PY'
import json
d=json.load(open("gpurun_out/r3c/bench_nocpu.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["ms_per_step"], d["roofline"].get("isolated"))
print({k:round(v["ms_per_step"],3) for k,v in d["roofline_table"].items()})
PY
