#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3l; mkdir -p $O
WM_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 > $O/bench_n2_shared.json 2> $O/bench_n2_shared.err; tail -2 $O/bench_n2_shared.err
python bench.py --no-cpu-baseline --no-train --no-bf16 --graph --steps 8 > $O/bench_graph.json 2> $O/bench_graph.err; tail -2 $O/bench_graph.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3l/bench_n2_shared.json")); print("n2", d["n_gpus"], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["data"])
d=json.load(open("gpurun_out/r3l/bench_graph.json")); print("graph", d["value"], d["hip_graph_replay"])
PY
