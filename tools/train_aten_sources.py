#!/usr/bin/env python3
"""Which lines of this package issue the small ATen kernels of a training step (BASELINE config 3 on one GPU)?  One profiled step with
Python stacks; the device kernels launched by aten:: operators are grouped by (operator, innermost frame inside wave_mamba_amd/)."""
import collections, os, sys
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    wm.trainer.train_step(net, opt, lq, gt, as_float=False)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if not e.name.startswith("aten::") or e.self_device_time_total <= 0:
        continue
    frame = "?"
    for fr in (e.stack or []):
        if "wave_mamba_amd" in fr or "bench.py" in fr:
            frame = fr.split("wave_mamba_amd/")[-1][:90]
            break
    if frame == "?" and e.stack:
        frame = e.stack[0][-90:]
    agg[(e.name, frame)][0] += 1
    agg[(e.name, frame)][1] += e.self_device_time_total
print(f"{'calls':>6} {'us':>9}  operator  <-  innermost frame in the package")
for (name, frame), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get('TOP', '70'))]:
    print(f"{n:6d} {t:9.0f}  {name:28s} <- {frame}")
