#!/bin/bash
O=gpurun_out/r5c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "core and (backward or bwd or grad)" 2>&1 | tail -3 | tee $O/tests.txt
timeout 300 python tools/bench_core_bwd.py 2>&1 | grep level | tee $O/bench_core_bwd.txt
WAVEMAMBA_HIP_LIB=build/variants/bwdstamp.so timeout 600 python tools/core_bwd_stamps.py 2>&1 | grep -v amdgpu | head -6 | tee $O/core_bwd_stamps.txt
