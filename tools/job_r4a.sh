#!/bin/bash
# round-4 first GPU job: the new size-of-record parity tests, the default bench line, the N = 2 shared-GPU self-launch
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4a; mkdir -p $O
timeout 1700 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "backward_at_training_sizes or wgrad_at_training or whole_model_gradients_hip" > $O/tests.txt 2>&1
tail -30 $O/tests.txt
timeout 900 python bench.py --steps 10 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
WM_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --train-steps 2 --no-bf16 > $O/bench_n2_shared.json 2> $O/bench_n2.err; echo "n2 rc $?"
tail -3 $O/bench_n2.err
