#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4q; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "lfss_prologue" 2>&1 | tail -8 | tee $O/tests.txt
python tools/bench_lfss_in.py 2>&1 | grep -v amdgpu | tee $O/bench_lfss_in.txt
