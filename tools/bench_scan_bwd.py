#!/usr/bin/env python3
"""Time selective_scan_fn forward + backward (HIP) at the BASELINE config-3 level shapes (B=8, KD=256)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wave_mamba_amd as wm
dev = "cuda:0"
for L in (65536, 16384, 4096):
    B, dim, N, G = 8, 256, 16, 4
    g = torch.Generator(device=dev).manual_seed(L)
    mk = lambda *s: torch.randn(*s, device=dev, generator=g)
    u, dl = mk(B, dim, L).requires_grad_(), (0.5 * mk(B, dim, L)).requires_grad_()
    A = (-torch.arange(1, N + 1, device=dev, dtype=torch.float32).repeat(dim, 1)).requires_grad_()
    Bm, Cm = mk(B, G, N, L).requires_grad_(), mk(B, G, N, L).requires_grad_()
    D, bias = torch.ones(dim, device=dev, requires_grad=True), torch.full((dim,), -4.0, device=dev, requires_grad=True)
    dy = mk(B, dim, L)
    def fb():
        y = wm.ops.selective_scan_fn(u, dl, A, Bm, Cm, D, None, bias, True)
        return torch.autograd.grad(y, (u, dl, A, Bm, Cm, D, bias), dy)
    fb(); torch.cuda.synchronize()
    wm.ops.prof_enable(True)
    for _ in range(3): fb()
    prof = wm.ops.prof_collect(); wm.ops.prof_enable(False)
    print(f"L={L}: " + "  ".join(f"{k} {v[1] / 3:.3f} ms" for k, v in prof.items() if v[0]))
