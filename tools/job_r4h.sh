#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4h; mkdir -p $O
echo "== only k=0 and k=1 chunk kernels (no accumulating launches)"; WM_CORE_BWD_DIRMASK=3 python tools/debug_core_bwd.py 8 64 256 256 16 2 2>&1 | grep -v amdgpu.ids | grep -A3 "fwd run" | tee $O/debug_mask3.txt
