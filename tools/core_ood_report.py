#!/usr/bin/env python3
"""Error table of the fused SS2D core (forward + backward) on the out-of-distribution cases of tests/test_gpu_parity.py
(ood_core_case): every tensor against the float64 evaluation, beside the fp32 reference arithmetic's own error.
   python tools/core_ood_report.py [--kinds trained dtpush ...]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T
from conftest import rel_err
from oracle import oracle
import wave_mamba_amd as wm

ap = argparse.ArgumentParser()
ap.add_argument("--kinds", nargs="*", default=["trained", "dtpush", "small", "large", "zeroplane", "zeroD"])
ap.add_argument("--shapes", type=int, nargs="*", default=[0, 1, 2, 3])
args = ap.parse_args()
names = ("dx", "dWx", "dWdt", "dbias", "dA_logs", "dDs")
for si in args.shapes:
    B, D, H, W, N, R = T.OOD_SHAPES[si]
    for kind in args.kinds:
        case = T.ood_core_case(B, D, H, W, N, R, seed=1000 + H * 7 + W + N, kind=kind)
        dys = [torch.randn(B, D, H * W, generator=T.gen(17 + i)) for i in range(4)]
        want = oracle.ss2d_core_raw(*case)
        ty, tg = T.core_eval(*case, dys, torch.float64)
        ry, rg = T.core_eval(*case, dys, torch.float32)
        a = [t.clone().requires_grad_(True) for t in T.cu(*case)]
        got = wm.ops.ss2d_core(*a)
        gr = torch.autograd.grad(got, a, T.cu(*dys))
        f = lambda x, y: max(rel_err(x.detach(), y))
        print(f"{(B, D, H, W, N, R)} {kind}:")
        print("   y  got/f64 " + " ".join(f"{f(g, t):.1e}" for g, t in zip(got, ty)) + "   oracle/f64 " +
              " ".join(f"{f(o, t):.1e}" for o, t in zip(want, ty)) + "   f32loop/f64 " + " ".join(f"{f(r, t):.1e}" for r, t in zip(ry, ty)))
        print("   grads got/f64 " + " ".join(f"{n} {f(g, t):.1e}" for n, g, t in zip(names, gr, tg)))
        print("        f32loop/f64 " + " ".join(f"{n} {f(g, t):.1e}" for n, g, t in zip(names, rg, tg)), flush=True)
