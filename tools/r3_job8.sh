#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3h; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -k "ss2d_core or lfss_block or core_abi or forward_is_bit" > $O/tests.log 2>&1; tail -4 $O/tests.log
python tools/bench_core.py --iters 5 > $O/core_rs2.log 2>&1
for v in rs1 rs3 rs4; do WAVEMAMBA_HIP_LIB=build/variants/$v.so python tools/bench_core.py --iters 5 > $O/core_$v.log 2>&1; done
python tools/bench_core.py --iters 3 --dstate 32 --levels 1 > $O/core_n32.log 2>&1
for l in 1 3; do WAVEMAMBA_HIP_LIB=build/variants/stamp2.so python tools/core_stamps.py --level $l > $O/stamps_l$l.log 2>&1; done
cat $O/core_*.log $O/stamps_*.log
