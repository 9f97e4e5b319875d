#!/bin/bash
# round-4 second GPU job: the second-generation fused-core backward - parity, then A/B timing against the first generation
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -x -k "ss2d_core_backward or trainable_lfss_block or backward_at_training_sizes" > $O/tests_core.txt 2>&1
tail -25 $O/tests_core.txt
timeout 600 python tools/train_breakdown.py --steps 3 --detail core_bwd_chunk,core_bwd_reduce > $O/train_v2.txt 2>&1; head -40 $O/train_v2.txt
WM_CORE_BWD_V1=1 timeout 600 python tools/train_breakdown.py --steps 3 > $O/train_v1.txt 2>&1; head -12 $O/train_v1.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "training_step or two_training_steps" > $O/tests_train.txt 2>&1
tail -12 $O/tests_train.txt
