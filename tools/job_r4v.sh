#!/bin/bash
O=gpurun_out/r4v; mkdir -p $O
timeout 900 python tools/train_aten_ops.py --top 400 2>&1 | grep -v "amdgpu\|Warn\|warn" > $O/train_aten_ops.txt
grep -v "^ .*ms .*\(void \|Cijk\|miopenSp3\|wm::\|Memset\)" $O/train_aten_ops.txt | cut -c1-200 | head -120
