#!/bin/bash
# round-3 GPU job 1: bf16x3 projection + prep kernel + planner: parity, accuracy, A/B timing
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3a; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -k "ss2d_core or lfss_block or core_abi" > $O/tests.log 2>&1; tail -5 $O/tests.log
python tools/core_accuracy.py > $O/acc_new.log 2>&1
WAVEMAMBA_HIP_AB=1 WAVEMAMBA_HIP_LIB=build/variants/r2.so python tools/core_accuracy.py > $O/acc_r2.log 2>&1
for v in r2 f32proj; do
  WAVEMAMBA_HIP_AB=1 WAVEMAMBA_HIP_LIB=build/variants/$v.so python tools/bench_core.py --iters 5 > $O/core_$v.log 2>&1
done
python tools/bench_core.py --iters 5 > $O/core_new.log 2>&1
for m in 5 10 1 2; do WM_CORE_DIRMASK=$m python tools/bench_core.py --iters 3 > $O/core_new_mask$m.log 2>&1; done
python tools/bench_core.py --iters 3 --dstate 32 --levels 1 > $O/core_new_n32.log 2>&1
WAVEMAMBA_HIP_AB=1 WAVEMAMBA_HIP_LIB=build/variants/r2.so python tools/bench_core.py --iters 3 --dstate 32 --levels 1 > $O/core_r2_n32.log 2>&1
cat $O/acc_*.log $O/core_*.log
