#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4m; mkdir -p $O
WM_TRAIN_CONV_BF16X3=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "shipped_config_256 or per_parameter_gradients or training_step_on_gpu_matches" 2>&1 | grep -v "^$\|Warning\|warn" | tail -15 | tee $O/fast_mode_tests.txt
WM_TRAIN_CONV_BF16X3=1 timeout 600 python tools/train_breakdown.py --steps 3 2>&1 | sed -n 3,16p | tee $O/train_fast.txt
