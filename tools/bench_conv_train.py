#!/usr/bin/env python3
"""Forward and input gradient of the dense convolutions at BASELINE config 3's shapes (batch 8): the fp16-split matrix-core
kernels (ops.conv2d_f16: amax + weight fragments + convolution) against ATen's fp32 (MIOpen Winograd / hipBLASLt), ms per call
and error against the float64 convolution."""
import os, sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wave_mamba_amd as wm
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
shapes = [(3, 64, 64, 256), (3, 64, 32, 256), (3, 32, 96, 256), (3, 64, 64, 128), (3, 64, 64, 64), (3, 3, 32, 512), (3, 32, 3, 512),
          (1, 64, 64, 256), (1, 32, 32, 256), (1, 32, 64, 256), (1, 32, 96, 256), (1, 64, 32, 128), (1, 32, 32, 64)]


def timeit(fn, n=10):
    """-> (host-clock ms per call of a back-to-back loop, GPU kernel ms per call from torch.profiler)"""
    from torch.profiler import profile, ProfilerActivity
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
    dev_us = sum(e.device_time for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA)
    return wall, dev_us / n / 1e3


for ks, ci, co, hw in shapes:
    x = torch.randn(8, ci, hw, hw, device=dev, generator=g)
    w = torch.randn(co, ci, ks, ks, device=dev, generator=g) / (ks * ci ** 0.5)
    t_hip = timeit(lambda: wm.ops.conv2d_f16(x, w))
    t_aten = timeit(lambda: F.conv2d(x, w, padding=ks // 2))
    xs = x[:2]
    ref = F.conv2d(xs.double(), w.double(), padding=ks // 2).float()
    e_hip = float((wm.ops.conv2d_f16(xs, w) - ref).norm() / ref.norm())
    e_aten = float((F.conv2d(xs, w, padding=ks // 2) - ref).norm() / ref.norm())
    print(f"{ks}x{ks} 8x{ci}->{co} {hw}x{hw}: fp16-split HIP {t_hip[1]:.3f} ms of kernels, {t_hip[0]:.3f} per call in a host loop ({e_hip:.1e})   "
          f"ATen fp32 {t_aten[1]:.3f} / {t_aten[0]:.3f} ms ({e_aten:.1e})")
