#!/usr/bin/env python3
"""Phase breakdown of core_bwd_chunk_kernel from in-kernel cycle stamps (library built with -DWM_BWD_STAMP=1):
   WAVEMAMBA_HIP_LIB=build/variants/bwdstamp.so python tools/core_bwd_stamps.py"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wave_mamba_amd as wm
from wave_mamba_amd import _lib
lib = ctypes.CDLL(_lib.LIB_PATH)
dev = "cuda:0"
names = ["inter-chunk barrier", "tile loads + barrier", "projection + dt", "barrier", "start state + forward sweep",
         "four sub-tiles in reverse", "barrier", "closing + barrier", "dx / dWx products + barrier", "dx store"]
for (B, D, H, W) in [(8, 64, 256, 256), (8, 64, 64, 64)]:
    N, R = 16, 2
    g = torch.Generator(device=dev); g.manual_seed(1)
    x = torch.randn(B, D, H, W, device=dev, generator=g).requires_grad_(True)
    ps = [torch.randn(4, R + 2 * N, D, device=dev, generator=g) / 8, torch.randn(4, D, R, device=dev, generator=g) * 0.7,
          torch.randn(4, D, device=dev, generator=g) * 0.5 - 3.0,
          torch.log(torch.arange(1, N + 1, device=dev, dtype=torch.float32)).repeat(4 * D, 1), torch.ones(4 * D, device=dev)]
    ps = [p.requires_grad_(True) for p in ps]
    buf = (ctypes.c_ulonglong * 44)()
    for it in range(3):
        ys = wm.ops.ss2d_core(x, *ps)
        ys = ys if isinstance(ys, (tuple, list)) else (ys,)
        gys = [torch.randn_like(y) for y in ys]
        torch.cuda.synchronize()
        lib.wm_debug_bwd_stamps(buf, 1)
        torch.autograd.backward(list(ys), gys)
        torch.cuda.synchronize()
    lib.wm_debug_bwd_stamps(buf, 0)
    v = list(buf)
    print(f"{B} x {D} x {H} x {W}: cycles per chunk (16 steps) and wave, summed over the launches of one backward call")
    for rev in range(2):
        for w in range(2):
            base = (rev * 2 + w) * 11
            n = max(v[base + 10], 1)
            tot = sum(v[base:base + 10]) / n
            print(f"  {'mirrored' if rev else 'forward '} direction kernels, wave {w}: {n} chunks, {tot:8.0f} cycles per chunk: " +
                  "  ".join(f"{nm} {v[base + k] / n:.0f}" for k, nm in enumerate(names)))
