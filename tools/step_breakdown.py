#!/usr/bin/env python3
"""Per-step kernel-time breakdown from a rocprofv3 --kernel-trace CSV of bench.py (steps delimited by the three
Haar analysis launches of a forward).  python tools/step_breakdown.py <bench_kernel_trace.csv> [nsteps]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'selscan_chunk_kernel' in r['Kernel_Name']]
rows = rows[:idx[0]] if idx else rows              # drop the op-boundary scan leg that follows the timed region
ha = [i for i, r in enumerate(rows) if 'haar_analysis' in r['Kernel_Name']]
starts = ha[0::3]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
# `bench.py --steps 5 --warmup 3 --timed-only`: 3 warm-up + 5 timed forwards; use timed steps 2..4 (the last one has no
# successor to delimit it)
s0, s1 = starts[3 + 1], starts[3 + 1 + n]
step = rows[s0:s1]
dur = lambda r: int(r['End_Timestamp']) - int(r['Start_Timestamp'])
agg = collections.defaultdict(lambda: [0, 0])
gaps = 0
for a, b in zip(step, step[1:]):
    g = int(b['Start_Timestamp']) - int(a['End_Timestamp'])
    if g > 0: gaps += g
for r in step:
    agg[r['Kernel_Name']][0] += 1; agg[r['Kernel_Name']][1] += dur(r)
tot = sum(v[1] for v in agg.values()); wmt = sum(v[1] for k, v in agg.items() if 'wm::' in k)
wall = (int(rows[s1]['Start_Timestamp']) - int(rows[s0]['Start_Timestamp'])) / n / 1e6
print(f"# per-step kernel time, mean of {n} timed steps of `rocprofv3 --kernel-trace -- python bench.py --steps 5 --warmup 3 --no-cpu-baseline --timed-only`")
print(f"# step wall {wall:.2f} ms; sum of kernel durations {tot / n / 1e6:.2f} ms/step in {len(step) / n:.0f} dispatches (kernels of the side "
      f"streams overlap the main stream's: the sum exceeds the wall, and a kernel sharing the GPU runs longer than alone); "
      f"wm:: kernels {wmt / n / 1e6:.2f} ms/step")
print("calls/step  ms/step  kernel")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:48]:
    print(f"{v[0] / n:9.1f} {v[1] / n / 1e6:8.3f}  {k[:130]}")
small = [(k, v) for k, v in agg.items() if v[1] / v[0] < 10000]
print(f"# dispatches shorter than 10 us: {sum(v[0] for _, v in small) / n:.0f} per step, {sum(v[1] for _, v in small) / n / 1e6:.3f} ms/step of kernel time")
