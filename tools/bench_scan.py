#!/usr/bin/env python3
"""Time the op-boundary selective scan (wm_selscan_fwd) at the three UHD pyramid levels.
   WAVEMAMBA_HIP_LIB=<variant.so> python tools/bench_scan.py [--iters 10]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wave_mamba_amd as wm

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--levels", type=int, nargs="*", default=[1, 2, 3])
ap.add_argument("--n", type=int, default=16)
args = ap.parse_args()
dev = "cuda:0"
print("lib:", wm._lib.LIB_PATH)
for lvl in args.levels:
    L = (2176 >> lvl) * (3840 >> lvl)
    dim, N, G = 256, args.n, 4
    g = torch.Generator(device=dev); g.manual_seed(lvl)
    u = torch.randn(1, dim, L, device=dev, generator=g)
    dl = 0.5 * torch.randn(1, dim, L, device=dev, generator=g)
    A = -torch.arange(1, N + 1, device=dev, dtype=torch.float32).repeat(dim, 1) * torch.exp(0.2 * torch.randn(dim, N, device=dev, generator=g))
    Bm = torch.randn(1, G, N, L, device=dev, generator=g); Cm = torch.randn(1, G, N, L, device=dev, generator=g)
    D = torch.randn(dim, device=dev, generator=g); bias = 0.5 * torch.randn(dim, device=dev, generator=g) - 4.0
    for _ in range(3):
        y = wm.ops.selective_scan_fn(u, dl, A, Bm, Cm, D, None, bias, True)
    torch.cuda.synchronize()
    wm.ops.prof_enable(True)
    for _ in range(args.iters):
        y = wm.ops.selective_scan_fn(u, dl, A, Bm, Cm, D, None, bias, True)
    prof = wm.ops.prof_collect(); wm.ops.prof_enable(False)
    r, c, s = (prof[k][1] / args.iters for k in ("selscan_chunk_reduce", "selscan_carry", "selscan_chunk_scan"))
    bytes_ = (4 * (3 * dim + 2 * G * N)) * L
    print(f"level {lvl} L={L:8d}: reduce {r:7.3f} ms  carry {c:6.3f} ms  scan {s:7.3f} ms  | scan-phase "
          f"{bytes_ / s / 1e6:7.1f} GB/s ({bytes_ / s / 1e6 / 8000:.3f} of 8 TB/s)  whole op {bytes_ / (r + c + s) / 1e6:7.1f} GB/s  "
          f"checksum {float(y.double().sum()):.6e}")
