#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4e; mkdir -p $O
python tools/debug_core_bwd.py 8 64 256 256 16 2 > $O/debug_l1.txt 2>&1; cat $O/debug_l1.txt
python tools/debug_core_bwd.py 2 64 64 64 16 2 > $O/debug_s.txt 2>&1; cat $O/debug_s.txt
