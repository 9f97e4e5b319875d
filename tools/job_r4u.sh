#!/bin/bash
# RCCL on the hardware: one-rank communicator - the GPU test, then bench.py's DDP leg under torch.distributed.run
O=gpurun_out/r4u; mkdir -p $O
timeout 900 python -m pytest tests/test_rccl_gpu.py -x -q -m gpu -s 2>&1 | grep -v "^\s*$" | tail -12 | tee $O/test_rccl.txt
WM_BENCH_RCCL_SELFTEST=1 NCCL_DEBUG=VERSION timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --train-steps 8 > $O/bench_rccl_one_rank.log 2>&1
grep -i "rccl\|nccl" $O/bench_rccl_one_rank.log | head -5
grep "^{" $O/bench_rccl_one_rank.log | tail -1 > $O/bench_rccl_one_rank.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4u/bench_rccl_one_rank.json"))
print("value", d["value"], d["unit"]); print("one gpu:", {k: d["training_config3_one_gpu"].get(k) for k in ("images_per_s", "ms_per_step")})
print("ddp:", {k: d["training_config3_ddp"].get(k) for k in ("images_per_s", "ms_per_step", "ms_per_step_without_allreduce", "exposed_allreduce_ms_per_step")})
PY
