#!/bin/bash
# rank-1 state update on the matrix pipe: microbenchmark, parity, A/B of the core at the UHD levels
O=gpurun_out/r4s; mkdir -p $O
timeout 300 ./tools/ubench_rank1_mfma 2>&1 | tee $O/ubench_rank1_mfma.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "core or ss2d or lfss" 2>&1 | tail -15 | tee $O/tests_core.txt
for i in 1 2; do
echo "== rank-1 on the matrix pipe (shipped)"; timeout 300 python tools/bench_core.py --iters 10 2>&1 | grep -v amdgpu | tee -a $O/bench_core.txt
echo "== packed FMA (variant base)"; WAVEMAMBA_HIP_LIB=build/variants/base.so timeout 300 python tools/bench_core.py --iters 10 2>&1 | grep -v amdgpu | tee -a $O/bench_core.txt
done
