#!/bin/bash
# PMC passes over the op-boundary selective scan at UHD level 1 (separate runs per counter group,
# as MI355X_MICROARCH.md prescribes).  Usage: tools/pmc_scan.sh <outdir> [lib.so]
set -u
R=$PWD; OUT=$R/$1; LIB=${2:-}
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
[ -n "$LIB" ] && export WAVEMAMBA_HIP_LIB=$R/$LIB
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python $R/tools/bench_scan.py --levels 1 --iters 3 > $OUT/$name.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
cd $R; find $OUT -name "*.csv" | head -20
