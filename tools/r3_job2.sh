#!/bin/bash
# round-3 GPU job 2: what bounds the core? active-CU / clock microbench, ablations, gradient errors vs the float64 truth
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3b; mkdir -p $O
tools/ubench_active_cus > $O/ubench_active_cus.txt 2>&1
for v in ablate1 ablate2 ablate4 ablate6; do
  WAVEMAMBA_HIP_AB=1 WAVEMAMBA_HIP_LIB=build/variants/$v.so python tools/bench_core.py --iters 5 > $O/core_$v.log 2>&1
done
python tools/bench_core.py --iters 5 > $O/core_new.log 2>&1
python tools/grad_deviation.py > $O/grad_dev.log 2>&1
WM_NO_HIP_CONV=1 python tools/grad_deviation.py > $O/grad_dev_noconv.log 2>&1
WM_NO_HIP_CORE=1 python tools/grad_deviation.py > $O/grad_dev_nocore.log 2>&1
cat $O/ubench_active_cus.txt $O/core_*.log $O/grad_dev*.log
