for i in 1 2 3; do timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --timed-only 2>&1 | tail -1 | cut -c1-200; done
timeout 800 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "multi_stream or pipeline or uhd_forward or batch" 2>&1 | tail -3
