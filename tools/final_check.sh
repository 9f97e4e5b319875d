#!/bin/bash
# complete GPU suite + smoke + round evidence + default bench on the current build
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/final; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x > $O/tests_full.log 2>&1; tail -3 $O/tests_full.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 2400 bash tools/profile_round.sh > $O/profile_round.log 2>&1; head -3 $O/profile_round.log
# the bench line quotes profiles/pmc_traffic.json only for the binary it describes: take the counters just collected on this build
cp gpurun_out/prof_round/pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null
timeout 900 bash tools/pmc_step_all.sh > $O/pmc_step_all.log 2>&1; tail -5 $O/pmc_step_all.log
timeout 600 python tools/host_bound.py > $O/host_bound.txt 2>&1; head -12 $O/host_bound.txt
timeout 1500 python bench.py 2> $O/bench_default.err | grep '^{' > $O/bench_default.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/final/bench_default.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"].get("traffic_over_algorithmic"), (d["roofline"].get("isolated") or {}).get("frac"))
print({k:(round(v["ms_per_step"],3), round(v.get("frac",0),3)) for k,v in d["roofline_table"].items()})
t=d["training_config3_one_gpu"]; print("train", t["images_per_s"], t["ms_per_step"]); print("hot", d["hot_path_sum"]["frac"], "concurrent", d["concurrent_forwards"]["images_per_s"], "bf16", d["bf16_storage"]["images_per_s"])
PY
