#!/bin/bash
# The per-round evidence on the shipped binary (run on the GPU box: gpurun -- bash tools/profile_round.sh): rocprofv3 kernel
# stats of the bench command, per-step breakdowns (multi-stream and single-stream), PMC traffic / busy fractions of the core
# forward, PMC + per-call timing of the fused-core backward (both generations), the training-step kernel table.
# Output under gpurun_out/prof_round/; copy what is to be kept into profiles/rNN/ and profiles/pmc_traffic.json.
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/prof_round; mkdir -p $O
python -c "import wave_mamba_amd as wm; print('build_id', wm._lib.build_id())" > $O/build_id.txt 2>&1
bash tools/profile_bench.sh $O/multi > $O/profile_multi.log 2>&1
WM_TWO_STREAMS=0 bash tools/profile_bench.sh $O/single > $O/profile_single.log 2>&1
bash tools/pmc_core.sh $O/pmc_core > $O/pmc_core.log 2>&1
python tools/pmc_traffic.py $O/pmc_core $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1
python tools/bench_core_bwd.py > $O/bench_core_bwd.txt 2>&1
bash tools/pmc_core_bwd.sh $O/pmc_core_bwd > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc_core_bwd core_bwd > $O/pmc_core_bwd_summary.txt 2>&1
python tools/train_breakdown.py --steps 3 --detail core_bwd_chunk,core_bwd_reduce > $O/train_step_kernel_breakdown.txt 2>&1
bash tools/pmc_conv.sh $O/pmc_conv > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc_conv conv > $O/pmc_conv_summary.txt 2>&1
python tools/bench_conv_train.py > $O/bench_conv_train.txt 2>&1
python tools/bench_lfss_rz.py > $O/bench_lfss_rz.txt 2>&1
cat $O/build_id.txt; cat $O/pmc_traffic.log; head -12 $O/multi/bench_rocprofv3_kernel_stats.csv; head -20 $O/single/bench_per_step_kernel_breakdown.txt; cat $O/bench_core_bwd.txt; ls -la $O $O/multi $O/single | head -40
