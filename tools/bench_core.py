#!/usr/bin/env python3
"""Time the fused SS2D core (wm_ss2d_core_fwd, merged) at the three UHD pyramid levels.
   WAVEMAMBA_HIP_LIB=<variant.so> python tools/bench_core.py [--iters 5] [--levels 1 2 3]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wave_mamba_amd as wm

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--levels", type=int, nargs="*", default=[1, 2, 3])
ap.add_argument("--merged", action="store_true", help="time the merged operator (adds the four-way sum kernel)")
ap.add_argument("--dstate", type=int, default=16)
args = ap.parse_args()
dev = "cuda:0"
print("lib:", wm._lib.LIB_PATH)
for lvl in args.levels:
    H, W = 2176 >> lvl, 3840 >> lvl
    D, N, R = 64, args.dstate, 2
    g = torch.Generator(device=dev); g.manual_seed(lvl)
    x = torch.randn(1, D, H, W, device=dev, generator=g)
    Wx = torch.randn(4, R + 2 * N, D, device=dev, generator=g) / 8
    Wdt = torch.randn(4, D, R, device=dev, generator=g) * 0.7
    bias = torch.randn(4, D, device=dev, generator=g) * 0.5 - 3.0
    A_logs = torch.log(torch.arange(1, N + 1, device=dev, dtype=torch.float32)).repeat(4 * D, 1)
    Ds = torch.ones(4 * D, device=dev)
    for _ in range(2):
        y = wm.ops.ss2d_core(x, Wx, Wdt, bias, A_logs, Ds, merged=args.merged)
    torch.cuda.synchronize()
    wm.ops.prof_enable(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        y = wm.ops.ss2d_core(x, Wx, Wdt, bias, A_logs, Ds, merged=args.merged)
    e1.record(); torch.cuda.synchronize()
    prof = wm.ops.prof_collect(); wm.ops.prof_enable(False)
    tot = e0.elapsed_time(e1) / args.iters
    parts = {k: v[1] / args.iters for k, v in prof.items() if v[0]}
    print(f"level {lvl} {H}x{W}: total {tot:7.3f} ms  " + "  ".join(f"{k} {v:.3f}" for k, v in parts.items())
          + f"  checksum {float((y if args.merged else sum(y)).double().sum()):.6e}")
