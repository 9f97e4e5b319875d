mkdir -p gpurun_out/r3z
python -m pytest tests -m gpu -x -q -k "grad or train or optimizer or conv2d_train" > gpurun_out/r3z/bwd_tests.log 2>&1; tail -3 gpurun_out/r3z/bwd_tests.log
python tools/train_breakdown.py > gpurun_out/r3z/train_breakdown.txt 2>&1; grep -v Warn gpurun_out/r3z/train_breakdown.txt | head -24
