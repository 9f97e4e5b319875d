#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4o; mkdir -p $O
python tools/train_module_breakdown.py 2>&1 | grep -v "amdgpu\|Warn\|warn" | tee $O/modules.txt
timeout 900 python -m pytest tests/test_torch_library_ops.py -m gpu -q 2>&1 | tail -3
