#!/usr/bin/env python3
"""Host-side cost of one operator call (microseconds), by piece: is a step bound by the CPU issuing it?
python tools/host_overhead.py   (on the GPU box)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wave_mamba_amd as wm
from wave_mamba_amd import _lib
dev = torch.device("cuda", 0)
x = torch.randn(1, 8, 4, 32, device=dev)
lib = _lib.load()
def t(f, n=2000):
    for _ in range(50): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    dt = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return dt
def ctx():
    with torch.cuda.device(dev): pass
print(f"with torch.cuda.device(dev): pass        {t(ctx):6.2f} us")
print(f"torch.cuda.current_stream().cuda_stream  {t(lambda: torch.cuda.current_stream().cuda_stream):6.2f} us")
print(f"torch.empty(1024, device)                {t(lambda: torch.empty(1024, device=dev)):6.2f} us")
print(f"x.contiguous().float() (no-ops)          {t(lambda: x.contiguous().float()):6.2f} us")
print(f"x.data_ptr()                             {t(lambda: x.data_ptr()):6.2f} us")
print(f"ctypes call wm_abi_version               {t(lambda: lib.wm_abi_version()):6.2f} us")
print(f"ops.plane_sums(x) whole call             {t(lambda: wm.ops.plane_sums(x)):6.2f} us")
print(f"torch x + x (ATen eager op)              {t(lambda: x + x):6.2f} us")
w = torch.randn(8, 8, 1, 1, device=dev, requires_grad=True)
gy = torch.randn(1, 8, 4, 32, device=dev)
def fb():
    y = wm.ops.conv2d_train(x, w, None)
    y.backward(gy)
print(f"conv2d_train fwd + bwd (1x1, tiny)       {t(fb, 500):6.2f} us")
def fb2():
    y = torch.nn.functional.conv2d(x, w)
    y.backward(gy)
print(f"F.conv2d fwd + bwd (1x1, tiny)           {t(fb2, 500):6.2f} us")
