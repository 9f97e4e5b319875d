#!/usr/bin/env python3
"""Forward + backward GPU time of single modules at BASELINE config 3's level-1 size (batch 8, 32 channels, 256 x 256), by kernel:
which PyTorch-side (at::native, Cijk, miopen) kernels each module still launches in training."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wave_mamba_amd as wm
from wave_mamba_amd.archs import wavemamba_arch as arch
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
torch.manual_seed(0)
B, C, H, W = 8, 32, 256, 256
x = torch.randn(B, C, H, W, device=dev, requires_grad=True)
p = torch.randn(B, C, H, W, device=dev, requires_grad=True)
bands = [torch.randn(B, C, H, W, device=dev, requires_grad=True) for _ in range(3)]
mods = {"HFEBlock": (arch.HFEBlock(C, match_factor=1, ffn_expansion_factor=1).to(dev).train(), lambda m: m(x, p)),
        "SKFF": (arch.SKFF(C, height=3, reduction=8).to(dev).train(), lambda m: m(bands)),
        "LFSSBlock": (arch.LFSSBlock(C, expand=2.0).to(dev).train(), lambda m: m.forward_nchw_train(x)),
        "l_conv 64->32 3x3": (torch.nn.Conv2d(64, 32, 3, 1, 1).to(dev).train(), lambda m: arch._conv(m, x, p))}
for name, (m, run) in mods.items():
    for _ in range(2):
        y = run(m); y.backward(torch.ones_like(y))
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        y = run(m); y.backward(torch.ones_like(y))
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            k = e.name.split("(")[0][:100]
            agg[k][0] += 1; agg[k][1] += e.device_time
    tot = sum(v[1] for v in agg.values())
    nat = sum(v[1] for k, v in agg.items() if "wm::" not in k)
    print(f"== {name}: {tot / 1e3:.3f} ms forward + backward in {sum(v[0] for v in agg.values())} kernels; not wm:: {nat / 1e3:.3f} ms")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
        print(f"   {v[0]:4d} {v[1] / 1e3:7.3f} ms  {k}")
