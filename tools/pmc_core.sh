#!/bin/bash
# PMC passes (separate runs per counter group, MI355X_MICROARCH.md) over the SS2D core at the three UHD pyramid levels,
# plus the float4 copy of tools/microbench (a known byte count: calibrates FETCH_SIZE / WRITE_SIZE on this box).
# Usage: tools/pmc_core.sh <outdir> [extra bench_core.py args]     (then: python tools/pmc_traffic.py <outdir>)
set -u
R=$PWD; OUT=$R/$1; shift
mkdir -p $OUT; export TMPDIR=/tmp
# the calibration copy is a built artefact (git-ignored): build it where it is missing (hipcc is in the image, here and on the GPU box)
[ -x $R/tools/microbench ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/microbench.hip -o $R/tools/microbench > $OUT/microbench_build.log 2>&1
cd /tmp
EXTRA="$*"
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python $R/tools/bench_core.py --levels 1 2 3 --iters 3 $EXTRA > $OUT/$name.log 2>&1; }
cal() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- $R/tools/microbench > $OUT/$name.log 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE
cal cal_fetch FETCH_SIZE
cal cal_write WRITE_SIZE
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE
cd $R
# keep the merged-back output small: only the rows of this library's kernels and of the calibration copy
for f in $OUT/*/p_counter_collection.csv; do
  (head -1 $f; grep -E '"void wm::|"wm::|copy_kernel' $f) > $f.tmp && mv $f.tmp $f
done
rm -f $OUT/*/p_kernel_trace.csv $OUT/*/p_agent_info.csv
