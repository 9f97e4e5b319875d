#!/usr/bin/env python3
"""A/B of the core's output planes at the three UHD levels: four planes (one per direction, lfss_mid adds them) against the
PAIRED mode (two planes: each reversed direction's launch adds into its forward twin's plane, lfss_mid reads two).
ms per call (HIP events): core alone, lfss_mid (gate recomputed) alone, the two back to back; and the error of the paired
planes against the four planes' pairwise sums."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wave_mamba_amd as wm
from wave_mamba_amd import _lib
from wave_mamba_amd.ops import _ptr, _stream, check

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--levels", type=int, nargs="*", default=[1, 2, 3])
args = ap.parse_args()
dev = "cuda:0"
lib = _lib.load()
C, D, N, R = 32, 64, 16, 2
print("lib:", _lib.LIB_PATH, "build", wm.build.source_id() if hasattr(wm, "build") else "")


def timed(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for lvl in args.levels:
    H, W = 2176 >> lvl, 3840 >> lvl
    L, B = H * W, 1
    g = torch.Generator(device=dev); g.manual_seed(lvl)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)
    x = rn(1, D, H, W)
    Wx, Wdt, bias = rn(4, R + 2 * N, D) / 8, rn(4, D, R) * 0.7, rn(4, D) * 0.5 - 3.0
    A_logs = torch.log(torch.arange(1, N + 1, device=dev, dtype=torch.float32)).repeat(4 * D, 1)
    Ds = torch.ones(4 * D, device=dev)
    ln1w, ln1b, ln2w, ln2b = rn(C) * 0.1 + 1, rn(C) * 0.1, rn(C) * 0.1 + 1, rn(C) * 0.1
    onw, onb = rn(D) * 0.1 + 1, rn(D) * 0.1
    Win, Wout, W1, b1, sk1 = rn(2 * D, C) / 6, rn(C, D) / 8, rn(D, C) / 6, rn(D) * 0.1, rn(C) * 0.1 + 1
    tok = rn(B, C, L)
    tok1, f = torch.empty(B, L, C, device=dev), torch.empty(B, D, L, device=dev)
    st = _stream()
    tail = (_ptr(tok), 1, _ptr(ln1w), _ptr(ln1b), 1e-5, _ptr(Win), _ptr(onw), _ptr(onb), 1e-5, _ptr(Wout), _ptr(sk1), _ptr(ln2w),
            _ptr(ln2b), 1e-5, _ptr(W1), _ptr(b1), _ptr(tok1), _ptr(f), B, L, C, 0, st)
    core = lambda m: wm.ops.ss2d_core(x, Wx, Wdt, bias, A_logs, Ds, merged=m)
    y4 = core(0); y2 = core(2)
    ref = (y4[0].double() + y4[1].double(), y4[2].double() + y4[3].double())
    err = max(float((a.double() - r).norm() / r.norm()) for a, r in zip(y2, ref))
    y4b = torch.stack(y4); y2b = torch.stack(y2)
    mid = lambda yb, ny: check(lib.wm_lfss_mid_rz_fwd(_ptr(yb), ny, B * D * L, *tail), "mid")
    mid(y4b, 4); f4 = f.clone(); mid(y2b, 2); f2 = f.clone()
    errf = float((f2.double() - f4.double()).norm() / f4.double().norm())
    t_c4, t_c2 = timed(lambda: core(0), args.iters), timed(lambda: core(2), args.iters)
    t_m4, t_m2 = timed(lambda: mid(y4b, 4), args.iters), timed(lambda: mid(y2b, 2), args.iters)
    t_b4 = timed(lambda: (core(0), mid(y4b, 4)), args.iters)
    t_b2 = timed(lambda: (core(2), mid(y2b, 2)), args.iters)
    print(f"level {lvl} {H}x{W}: core {t_c4:.3f} -> {t_c2:.3f} ms, lfss_mid {t_m4:.3f} -> {t_m2:.3f} ms, "
          f"core + lfss_mid {t_b4:.3f} -> {t_b2:.3f} ms (four planes -> paired); paired planes vs pairwise sums rel {err:.2e}, "
          f"lfss_mid f rel {errf:.2e}")
