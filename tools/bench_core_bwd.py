#!/usr/bin/env python3
"""wm_ss2d_core_bwd alone at BASELINE config 3's three pyramid levels (batch 8: 256 x 256, 128 x 128, 64 x 64 maps; merged
output gradient, as LFSSBlock's y1 + y2 + y3 + y4 produces it): ms per call from HIP events around the whole backward, and
- under `rocprofv3 --pmc` (tools/pmc_core_bwd.sh) - the counters of its kernels.
    python tools/bench_core_bwd.py [--levels 1 2 3] [--iters 5]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wave_mamba_amd as wm
ap = argparse.ArgumentParser()
ap.add_argument("--levels", type=int, nargs="+", default=[1, 2, 3])
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--size", type=int, default=512)
args = ap.parse_args()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
D, N, R = 64, 16, 2
Wx = (torch.randn(4, R + 2 * N, D, device=dev, generator=g) / 8).requires_grad_(True)
Wdt = (torch.randn(4, D, R, device=dev, generator=g) * 0.7).requires_grad_(True)
bias = (torch.randn(4, D, device=dev, generator=g) * 0.5 - 3.0).requires_grad_(True)
A_logs = (torch.log(torch.arange(1, N + 1, dtype=torch.float32, device=dev)).repeat(4 * D, 1)).requires_grad_(True)
Ds = torch.ones(4 * D, device=dev, requires_grad=True)
for lvl in args.levels:
    H = W = args.size >> lvl
    x = torch.randn(args.batch, D, H, W, device=dev, generator=g).requires_grad_(True)
    dy = torch.randn(args.batch, D, H * W, device=dev, generator=g)
    params = [x, Wx, Wdt, bias, A_logs, Ds]
    y = wm.ops.ss2d_core(*params, merged=True)
    torch.autograd.grad(y, params, dy, retain_graph=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        torch.autograd.grad(y, params, dy, retain_graph=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    pos = args.batch * H * W
    print(f"level {lvl}: {args.batch} x {D} x {H} x {W}: backward {ms:.3f} ms per call = {6144 * pos / ms / 1e6:.0f} GB/s on 6144 B / position "
          f"({6144 * pos / ms / 1e6 / 8000:.3f} of 8 TB/s)")
