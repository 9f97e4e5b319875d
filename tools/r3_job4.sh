#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3d; mkdir -p $O
python tools/grad_localize.py > $O/grad_localize.log 2>&1
WAVEMAMBA_HIP_LIB=build/variants/f32proj.so python tools/grad_localize.py > $O/grad_localize_f32proj.log 2>&1
python -m pytest tests -m gpu -q -k "training_step or backward" 2>&1 | tail -30 > $O/tests_grad.log
python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_parity.py::test_training_step_per_parameter_gradients_on_gpu --deselect tests/test_gpu_parity.py::test_training_step_on_gpu_matches_reference 2>&1 | tail -25 > $O/tests.log
cat $O/grad_localize.log $O/grad_localize_f32proj.log $O/tests_grad.log $O/tests.log
