#!/usr/bin/env python3
"""One training step (BASELINE config 3 on one GPU) by ATen / autograd operator: GPU time per op name (torch.profiler
key_averages), to see which PyTorch-side operators are left between the HIP kernels."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wave_mamba_amd as wm
import bench
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = wm.WaveMamba(**bench.SHIPPED).train().to(dev)
opt = wm.trainer.make_optimizer(net)
g = torch.Generator().manual_seed(1234)
lq, gt = torch.rand(8, 3, 512, 512, generator=g).to(dev), torch.rand(8, 3, 512, 512, generator=g).to(dev)
for _ in range(3):
    wm.trainer.train_step(net, opt, lq, gt, as_float=False)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    wm.trainer.train_step(net, opt, lq, gt, as_float=False)
    torch.cuda.synchronize()
rows = [(e.self_device_time_total, e.count, e.key) for e in prof.key_averages() if e.self_device_time_total > 0]
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"GPU time by operator, one step: {tot / 1e3:.2f} ms")
for t, n, k in rows[:45]:
    print(f"{t / 1e3:8.3f} ms {n:5d}  {k[:100]}")
