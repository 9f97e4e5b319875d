#!/bin/bash
O=gpurun_out/r5b; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "layernorm or dwconv or training_step or whole_model or grad" 2>&1 | tail -3 | tee $O/tests.txt
timeout 600 python tools/train_breakdown.py 2>&1 | grep -v "amdgpu\|Warn\|warn" | head -14 | tee $O/train.txt
