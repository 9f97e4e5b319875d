mkdir -p gpurun_out/r3z
python -m pytest tests -m gpu -x -q > gpurun_out/r3z/gpu_tests.log 2>&1; tail -3 gpurun_out/r3z/gpu_tests.log
python bench.py --no-cpu-baseline --no-train --no-bf16 > gpurun_out/r3z/bench_fast.json 2> gpurun_out/r3z/bench_fast.err; cat gpurun_out/r3z/bench_fast.json | cut -c1-600
python tools/train_breakdown.py --detail projgrad_kernel,selscan_bwd_chunk,selscan_bwd_reduce > gpurun_out/r3z/train_breakdown.txt 2>&1; grep "^#\|wall" gpurun_out/r3z/train_breakdown.txt
