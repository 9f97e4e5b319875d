#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4c; mkdir -p $O
python tools/bench_core_bwd.py > $O/bench_core_bwd_v2.txt 2>&1; cat $O/bench_core_bwd_v2.txt
WM_CORE_BWD_V1=1 python tools/bench_core_bwd.py > $O/bench_core_bwd_v1.txt 2>&1; cat $O/bench_core_bwd_v1.txt
bash tools/pmc_core_bwd.sh $O/pmc_v2 > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc_v2 core_bwd > $O/pmc_v2_summary.txt 2>&1; head -60 $O/pmc_v2_summary.txt
WM_CORE_BWD_V1=1 bash tools/pmc_core_bwd.sh $O/pmc_v1 > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc_v1 selscan_bwd > $O/pmc_v1_summary.txt 2>&1
timeout 600 python tools/train_ops.py > $O/train_ops.txt 2>&1; head -50 $O/train_ops.txt
