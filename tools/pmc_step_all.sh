#!/bin/bash
# PMC (five separate passes) over EVERY library kernel of the UHD inference step -> per-kernel traffic / busy table
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/pmc_step; rm -rf $O; mkdir -p $O
python -c "import wave_mamba_amd as wm; print('build_id', wm._lib.build_id())" > $O/build_id.txt 2>&1
WM_TWO_STREAMS=0 bash tools/pmc_step_kernel.sh $O 'wm::' > $O/run.log 2>&1
(cat $O/build_id.txt; python tools/pmc_step_table.py $O) > $O/pmc_step_table.txt 2>&1; cat $O/pmc_step_table.txt
# (the merged-back output stays small: counter rows of this library's kernels only, no traces)
for f in $O/*/p_counter_collection.csv; do (head -1 $f; grep -E '"void wm::|"wm::' $f) > $f.tmp && mv $f.tmp $f; done
rm -f $O/*/p_kernel_trace.csv $O/*/p_agent_info.csv; rm -rf $O/*/*.db
