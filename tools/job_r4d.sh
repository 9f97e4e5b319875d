#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4d; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -x -k "ss2d_core_backward or trainable_lfss_block or backward_at_training_sizes" > $O/tests_core.txt 2>&1
tail -8 $O/tests_core.txt
python tools/bench_core_bwd.py > $O/bench_core_bwd_v2.txt 2>&1; cat $O/bench_core_bwd_v2.txt
timeout 600 python tools/train_breakdown.py --steps 3 > $O/train_v2.txt 2>&1; head -24 $O/train_v2.txt
