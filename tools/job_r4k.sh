#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4k; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "gate_and_glu or training_step or two_training_steps or hfe_block or shipped_config_256 or whole_model" > $O/tests.txt 2>&1; grep -v "^$" $O/tests.txt | tail -12
timeout 600 python tools/train_breakdown.py --steps 3 > $O/train.txt 2>&1; sed -n 3,40p $O/train.txt
