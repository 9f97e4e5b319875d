#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4g; mkdir -p $O
python tools/debug_core_bwd.py 8 64 256 256 16 2 2>&1 | grep -v amdgpu.ids | grep -A4 "row fwd run 0\|col fwd run 0" | tee $O/debug_l1.txt
echo == 1 x 256 x 256; python tools/debug_core_bwd.py 1 64 256 256 16 2 2>&1 | grep -v amdgpu.ids | grep -A4 "row fwd run 0" | tee $O/debug_b1.txt
echo == 8 x 128 x 128; python tools/debug_core_bwd.py 8 64 128 128 16 2 2>&1 | grep -v amdgpu.ids | grep -A4 "row fwd run 0" | tee $O/debug_l2.txt
