#!/usr/bin/env python3
"""Per-kernel GPU time of one steady-state training step (BASELINE config 3 on one GPU: batch 8 x 3 x 512 x 512), from
torch.profiler.   python tools/train_breakdown.py [--steps 3]"""
import argparse, os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wave_mamba_amd as wm
import bench
ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=3); ap.add_argument("--wall-only", action="store_true")
ap.add_argument("--top", type=int, default=45); ap.add_argument("--ops", action="store_true", help="also: host-side operators by the GPU time of their own kernels")
ap.add_argument("--detail", default="", help="comma-separated kernel-name substrings: print every launch's duration (us) of the last profiled step")
args = ap.parse_args()
# (torch.backends.cudnn.benchmark = True, which the reference's train.py:129 sets, is NOT an option here: without a
# find-db MIOpen's exhaustive search compiles and times every solver per shape - the three warm-up steps did not finish in
# 25 minutes on the GPU box, gpurun_out r3z.)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = wm.WaveMamba(**bench.SHIPPED).train().to(dev)
opt = wm.trainer.make_optimizer(net)
g = torch.Generator().manual_seed(1234)
lq, gt = torch.rand(8, 3, 512, 512, generator=g).to(dev), torch.rand(8, 3, 512, 512, generator=g).to(dev)
import time
t0 = time.perf_counter()
for _ in range(3):
    wm.trainer.train_step(net, opt, lq, gt, as_float=False)
torch.cuda.synchronize()
print(f"three warm-up steps: {time.perf_counter() - t0:.1f} s")
t0 = time.perf_counter()
for _ in range(5):
    wm.trainer.train_step(net, opt, lq, gt, as_float=False)
torch.cuda.synchronize()
print(f"wall clock, un-profiled: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per step")
if "--wall-only" in sys.argv:
    sys.exit(0)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA] + ([ProfilerActivity.CPU] if args.ops else [])) as prof:
    for _ in range(args.steps):
        wm.trainer.train_step(net, opt, lq, gt, as_float=False)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        k = (e.name if e.name.startswith("void at::native") else e.name.split("(")[0])[:200 if e.name.startswith("void at::native") else 110]
        agg[k][0] += 1; agg[k][1] += e.device_time
if args.detail:
    evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    for sub in args.detail.split(","):
        d = [e.device_time for e in evs if sub in e.name]
        d = d[len(d) * (args.steps - 1) // args.steps:]
        print(f"# {sub}: {len(d)} launches in the last step, us: " + " ".join(f"{x:.0f}" for x in d))
tot = sum(v[1] for v in agg.values())
print(f"GPU kernel time per step: {tot / args.steps / 1e3:.2f} ms in {sum(v[0] for v in agg.values()) / args.steps:.0f} kernels")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:args.top]:
    print(f"{v[0] / args.steps:7.1f} {v[1] / args.steps / 1e3:8.3f} ms  {k}")
if args.ops:
    print("# host-side operators by the GPU time of the kernels they launch themselves (per step)")
    rows = sorted(prof.key_averages(), key=lambda a: -a.self_device_time_total)[:60]
    for a in rows:
        if a.self_device_time_total > 0:
            print(f"{a.count / args.steps:7.1f} {a.self_device_time_total / args.steps / 1e3:8.3f} ms  {a.key[:120]}")
