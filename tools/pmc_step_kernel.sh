#!/bin/bash
# PMC passes (separate runs per counter group) over ONE kernel class of the UHD inference step.  Usage: tools/pmc_step_kernel.sh <outdir> <kernel-substring>
set -u
R=$PWD; OUT=$R/$1; K=$2; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "$K" --output-format csv -d $OUT/$name -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --timed-only > $OUT/$name.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS
run sq3 SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM
run fetch FETCH_SIZE
run write WRITE_SIZE
cd $R; python tools/pmc_summary.py $OUT "$K" > $OUT/summary.txt; cat $OUT/summary.txt
