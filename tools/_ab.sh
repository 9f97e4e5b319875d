for ts in 1 0 1; do echo "== WM_TWO_STREAMS=$ts"; WM_TWO_STREAMS=$ts timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-260; done
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "uhd or network or batch or pipeline or drop" 2>&1 | tail -2
