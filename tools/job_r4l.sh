#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4l; mkdir -p $O
python tools/bench_conv_aten.py 2>&1 | grep -v amdgpu | tee $O/bench_conv_aten.txt
bash tools/pmc_core_bwd.sh $O/pmc_bwd > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc_bwd core_bwd > $O/pmc_bwd_summary.txt 2>&1
