#!/usr/bin/env python3
"""Time the three LFSS glue kernels (wm_lfss_in / mid / out) at the UHD pyramid levels against their HBM floors.
   WAVEMAMBA_HIP_LIB=<variant.so> python tools/bench_lfss.py [--iters 5] [--levels 1 2 3]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wave_mamba_amd as wm
from wave_mamba_amd import _lib
from wave_mamba_amd.ops import _ptr, _stream, check

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--levels", type=int, nargs="*", default=[1, 2, 3])
args = ap.parse_args()
dev = "cuda:0"
lib = _lib.load()
print("lib:", _lib.LIB_PATH)
C, D = 32, 64
g = torch.Generator(device=dev); g.manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
ln1w, ln1b, ln2w, ln2b = rn(C) * 0.1 + 1, rn(C) * 0.1, rn(C) * 0.1 + 1, rn(C) * 0.1
onw, onb = rn(D) * 0.1 + 1, rn(D) * 0.1
Win, Wout, W1, b1, W3, b3 = rn(2 * D, C) / 6, rn(C, D) / 8, rn(D, C) / 6, rn(D) * 0.1, rn(C, C) / 6, rn(C) * 0.1
sk1, sk2 = rn(C) * 0.1 + 1, rn(C) * 0.1 + 1


def timed(fn):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.iters


for lvl in args.levels:
    H, W = 2176 >> lvl, 3840 >> lvl
    L, B = H * W, 1
    tok = rn(B, L, C)
    x, z = torch.empty(B, D, L, device=dev), torch.empty(B, D, L, device=dev)
    ysum, fc = rn(B, D, L), rn(B, D, L)
    tok1, f, out = torch.empty(B, L, C, device=dev), torch.empty(B, D, L, device=dev), torch.empty(B, L, C, device=dev)
    st = _stream()
    t_in = timed(lambda: check(lib.wm_lfss_in_fwd(_ptr(tok), 0, _ptr(ln1w), _ptr(ln1b), 1e-5, _ptr(Win), _ptr(x), _ptr(z),
                                                  B, L, C, 0, st), "in"))
    t_mid = timed(lambda: check(lib.wm_lfss_mid_fwd(_ptr(ysum), 1, 0, _ptr(z), _ptr(tok), 0, _ptr(onw), _ptr(onb), 1e-5, _ptr(Wout),
                                                    _ptr(sk1), _ptr(ln2w), _ptr(ln2b), 1e-5, _ptr(W1), _ptr(b1), _ptr(tok1),
                                                    _ptr(f), B, L, C, 0, st), "mid"))
    y4 = rn(4, B, D, L)
    t_mid4 = timed(lambda: check(lib.wm_lfss_mid_fwd(_ptr(y4), 4, B * D * L, _ptr(z), _ptr(tok), 0, _ptr(onw), _ptr(onb), 1e-5, _ptr(Wout),
                                                     _ptr(sk1), _ptr(ln2w), _ptr(ln2b), 1e-5, _ptr(W1), _ptr(b1), _ptr(tok1),
                                                     _ptr(f), B, L, C, 0, st), "mid4"))
    t_sum = timed(lambda: y4[0].add_(y4[1]).add_(y4[2]).add_(y4[3]))
    t_out = timed(lambda: check(lib.wm_lfss_out_fwd(_ptr(fc), _ptr(tok1), _ptr(W3), _ptr(b3), _ptr(sk2), _ptr(out), 0,
                                                    B, L, C, 0, st), "out"))
    fl = lambda bytes_pp: bytes_pp * L / 5.0e9          # ms at the 5 TB/s copy ceiling
    print(f"level {lvl} {H}x{W}: in {t_in:.3f} ms (floor {fl(640):.3f})  mid {t_mid:.3f} ms (floor {fl(1024):.3f})  mid(4y) {t_mid4:.3f} (floor {fl(1792):.3f}; torch 3 adds {t_sum:.3f})  "
          f"out {t_out:.3f} ms (floor {fl(512):.3f})  checksums {float(x.double().sum()):.6e} {float(z.double().sum()):.6e} "
          f"{float(tok1.double().sum()):.6e} {float(f.double().sum()):.6e} {float(out.double().sum()):.6e}")
