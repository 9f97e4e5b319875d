#!/bin/bash
O=gpurun_out/r4w; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gram or training_step or whole_model or grad or hfe" 2>&1 | tail -15 | tee $O/tests.txt
timeout 600 python tools/train_breakdown.py 2>&1 | grep -v "amdgpu\|Warn\|warn" | head -40 | tee $O/train.txt
