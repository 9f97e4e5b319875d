#!/bin/bash
# PMC passes (separate runs per counter group, MI355X_MICROARCH.md) over the fused-core BACKWARD at BASELINE config 3's three
# pyramid levels.  Usage: tools/pmc_core_bwd.sh <outdir>   (then: python tools/pmc_summary.py <outdir> core_bwd)
set -u
R=$PWD; OUT=$R/$1; shift
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python $R/tools/bench_core_bwd.py --iters 2 > $OUT/$name.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE
run sq3 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
run fetch FETCH_SIZE
run write WRITE_SIZE
cd $R
for f in $OUT/*/p_counter_collection.csv; do
  (head -1 $f; grep -E '"void wm::|"wm::' $f) > $f.tmp && mv $f.tmp $f
done
rm -f $OUT/*/p_kernel_trace.csv $OUT/*/p_agent_info.csv
