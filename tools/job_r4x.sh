#!/bin/bash
O=gpurun_out/r4x; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "training_step or whole_model or grad" 2>&1 | grep -i "worst\|passed\|failed\|error" | tail -30 | tee $O/tests_train_auto.txt
for m in auto; do
echo "== WM_TRAIN_CONV=$m"; WM_TRAIN_CONV=$m timeout 600 python tools/train_breakdown.py 2>&1 | grep -v "amdgpu\|Warn\|warn" | head -40 | tee $O/train_$m.txt
done
