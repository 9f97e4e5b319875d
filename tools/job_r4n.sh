#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4n; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_torch_library_ops.py -m gpu -q -x -k "ss2d_core_backward or trainable_lfss_block or backward_at_training_sizes or library_ops" > $O/tests.txt 2>&1; tail -5 $O/tests.txt
python tools/bench_core_bwd.py 2>&1 | grep level | tee $O/bench_core_bwd.txt
timeout 600 python tools/train_breakdown.py --steps 3 --detail core_bwd_chunk 2>&1 | sed -n 4,12p | tee $O/train.txt
