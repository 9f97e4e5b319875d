#!/bin/bash
O=gpurun_out/r5i; mkdir -p $O; rm -f $O/ab.txt
for v in "" build/variants/mid2.so build/variants/mid4.so "" build/variants/mid2.so; do
  export WAVEMAMBA_HIP_LIB=$v; [ -z "$v" ] && unset WAVEMAMBA_HIP_LIB
  echo "== ${v:-shipped (3 waves)}" | tee -a $O/ab.txt
  timeout 300 python tools/bench_lfss_rz.py 2>&1 | grep level | cut -c1-150 | tee -a $O/ab.txt
done
