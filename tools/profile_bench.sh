#!/bin/bash
# rocprofv3 kernel trace + stats of the default bench command (GPU box).  Usage: tools/profile_bench.sh <outdir>
set -u
R=$PWD; OUT=$R/$1; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --timed-only > $OUT/bench_line.json 2> $OUT/bench.err
cd $R
python tools/step_breakdown.py $OUT/trace/bench_kernel_trace.csv 3 > $OUT/bench_per_step_kernel_breakdown.txt
python - "$OUT" <<'PY'
import csv, sys, collections
out = sys.argv[1]
rows = list(csv.DictReader(open(out + "/trace/bench_kernel_trace.csv")))
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    k = (r["Kernel_Name"].split("(")[0][:90], r["Grid_Size_X"], r["Workgroup_Size_X"])
    agg[k][0] += 1; agg[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
with open(out + "/bench_kernels_by_grid.txt", "w") as f:
    f.write("# whole run (warm-up + timed + untimed profiling pass): calls, total ms, avg us, kernel, grid x, block x\n")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:80]:
        f.write(f"{v[0]:6d} {v[1] / 1e6:9.3f} {v[1] / v[0] / 1e3:9.1f}  {k[0]}  grid {k[1]} block {k[2]}\n")
PY
cp $OUT/trace/bench_kernel_stats.csv $OUT/bench_rocprofv3_kernel_stats.csv 2>/dev/null
rm -rf $OUT/trace
