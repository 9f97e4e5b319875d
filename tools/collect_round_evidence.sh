#!/bin/bash
# Copy what tools/profile_round.sh + the default bench left under gpurun_out/ into profiles/<round>/ (build container, after the gpurun call).
# Usage: tools/collect_round_evidence.sh r04
set -e
cd "$(dirname "$0")/.."; P=gpurun_out/prof_round; R=profiles/$1; mkdir -p $R
cp $P/multi/bench_rocprofv3_kernel_stats.csv $R/bench_rocprofv3_kernel_stats.csv
cp $P/multi/bench_line.json $R/bench_profiled_run_line.json
cp $P/multi/bench_per_step_kernel_breakdown.txt $R/bench_per_step_kernel_breakdown.txt
cp $P/multi/bench_kernels_by_grid.txt $R/bench_kernels_by_grid.txt
cp $P/single/bench_per_step_kernel_breakdown.txt $R/bench_per_step_kernel_breakdown_single_stream.txt
cp $P/pmc_traffic.json $R/pmc_traffic.json; cp $P/pmc_traffic.json profiles/pmc_traffic.json
for d in pmc_core pmc_core_bwd; do
  rm -rf $R/$d; mkdir -p $R/$d
  find $P/$d -name "*counter_collection.csv" | while read f; do n=$(echo $f | sed "s#$P/$d/##; s#/#_#g"); cp $f $R/$d/$n; done
done
cp $P/bench_core_bwd.txt $R/core_bwd_per_call.txt
cp $P/pmc_core_bwd_summary.txt $R/pmc_core_bwd_summary.txt
cp $P/train_step_kernel_breakdown.txt $R/train_step_kernel_breakdown.txt
cp $P/pmc_conv_summary.txt $R/pmc_conv_summary.txt; cp $P/bench_conv_train.txt $R/bench_conv_train.txt; cp $P/bench_lfss_rz.txt $R/bench_lfss_recomputed_gate.txt
cp gpurun_out/final/bench_default.json $R/bench_default_line.json
cp gpurun_out/pmc_step/pmc_step_table.txt $R/pmc_step_traffic_per_kernel.txt; cp gpurun_out/final/host_bound.txt $R/host_bound.txt
tail -3 gpurun_out/final/tests_full.log > $R/tests_gpu_final_build.txt; tail -1 gpurun_out/final/smoke.log >> $R/tests_gpu_final_build.txt
cat $P/build_id.txt; python -c "
import sys; sys.path.insert(0, '.')
from wave_mamba_amd import build; print('local source id', build.source_id())"
