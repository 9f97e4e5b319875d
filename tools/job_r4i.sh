#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4i; mkdir -p $O
python tools/debug_core_bwd.py 8 64 256 256 16 2 2>&1 | grep -v amdgpu.ids | grep -A1 " run " | tee $O/debug.txt
