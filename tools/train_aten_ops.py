#!/usr/bin/env python3
"""Which PyTorch-side operators still cost GPU time in a config-3 training step: torch.profiler (CPU + CUDA activities), self
device time per operator name and input shapes, operators whose kernels are not wm:: only.
   python tools/train_aten_ops.py [--top 60]"""
import argparse, os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wave_mamba_amd as wm
import bench
from torch.profiler import profile, ProfilerActivity
ap = argparse.ArgumentParser(); ap.add_argument("--top", type=int, default=60); args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = wm.WaveMamba(**bench.SHIPPED).train().to(dev)
opt = wm.trainer.make_optimizer(net)
g = torch.Generator().manual_seed(1234)
lq, gt = torch.rand(8, 3, 512, 512, generator=g).to(dev), torch.rand(8, 3, 512, 512, generator=g).to(dev)
for _ in range(3):
    wm.trainer.train_step(net, opt, lq, gt, as_float=False)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    wm.trainer.train_step(net, opt, lq, gt, as_float=False)
    torch.cuda.synchronize()
rows = prof.key_averages(group_by_input_shape=True)
tot = 0.0
out = []
for r in rows:
    t = getattr(r, "self_device_time_total", 0.0)
    if t <= 0:
        continue
    tot += t
    out.append((t, r.count, r.key, str(r.input_shapes)[:150]))
out.sort(reverse=True)
print(f"self device time of all operators: {tot / 1e3:.2f} ms")
byname = collections.defaultdict(lambda: [0.0, 0])
for t, c, k, s in out:
    byname[k][0] += t; byname[k][1] += c
print("== by operator")
for k, (t, c) in sorted(byname.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"{t / 1e3:8.3f} ms {c:5d}  {k}")
print("== by operator and input shapes")
for t, c, k, s in out[:args.top]:
    print(f"{t / 1e3:8.3f} ms {c:4d}  {k:45s} {s}")
