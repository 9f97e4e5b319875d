#!/bin/bash
O=gpurun_out/r4v; mkdir -p $O
timeout 900 python tools/train_aten_ops.py --top 80 2>&1 | grep -v "amdgpu\|Warn\|warn" | tee $O/train_aten_ops.txt | head -150
