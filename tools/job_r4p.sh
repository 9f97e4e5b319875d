#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_torch_library_ops.py -m gpu -q 2>&1 | grep -E "^E |Error|error" | head -20
