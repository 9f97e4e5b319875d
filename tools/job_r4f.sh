#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4f; mkdir -p $O
echo "== default (nop, direct add)"; python tools/debug_core_bwd.py 8 64 256 256 16 2 2>&1 | grep -v amdgpu.ids | tee $O/debug_default.txt
echo "== late add, no nop"; WAVEMAMBA_HIP_LIB=$PWD/build/variants/late_nonop.so python tools/debug_core_bwd.py 8 64 256 256 16 2 2>&1 | grep -v amdgpu.ids | tee $O/debug_late_nonop.txt
echo "== late add, nop"; WAVEMAMBA_HIP_LIB=$PWD/build/variants/late_nop.so python tools/debug_core_bwd.py 8 64 256 256 16 2 2>&1 | grep -v amdgpu.ids | tee $O/debug_late_nop.txt
