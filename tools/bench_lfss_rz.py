#!/usr/bin/env python3
"""lfss_in + lfss_mid at the UHD levels: the gate z written by lfss_in and read back by lfss_mid against z recomputed inside
lfss_mid (wm_lfss_mid_rz_fwd, lfss_in writes the x half only).  ms per call (HIP events), ny = 4 as in the inference path."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wave_mamba_amd as wm
from wave_mamba_amd import _lib
from wave_mamba_amd.ops import _ptr, _stream, check
dev = "cuda:0"
lib = _lib.load()
C, D = 32, 64
g = torch.Generator(device=dev); g.manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
ln1w, ln1b, ln2w, ln2b = rn(C) * 0.1 + 1, rn(C) * 0.1, rn(C) * 0.1 + 1, rn(C) * 0.1
onw, onb = rn(D) * 0.1 + 1, rn(D) * 0.1
Win, Wout, W1, b1 = rn(2 * D, C) / 6, rn(C, D) / 8, rn(D, C) / 6, rn(D) * 0.1
sk1 = rn(C) * 0.1 + 1


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for lvl in (1, 2, 3):
    H, W = 2176 >> lvl, 3840 >> lvl
    L, B = H * W, 1
    tok = rn(B, C, L)                                     # NCHW tokens, as the first block of a stack gets them
    x, z = torch.empty(B, D, L, device=dev), torch.empty(B, D, L, device=dev)
    y4 = rn(4, B, D, L)
    tok1, f = torch.empty(B, L, C, device=dev), torch.empty(B, D, L, device=dev)
    st = _stream()
    args_tail = (_ptr(onw), _ptr(onb), 1e-5, _ptr(Wout), _ptr(sk1), _ptr(ln2w), _ptr(ln2b), 1e-5, _ptr(W1), _ptr(b1), _ptr(tok1), _ptr(f), B, L, C, 0, st)
    t_in = timed(lambda: check(lib.wm_lfss_in_fwd(_ptr(tok), 1, _ptr(ln1w), _ptr(ln1b), 1e-5, _ptr(Win), _ptr(x), _ptr(z), B, L, C, 0, st), "in"))
    t_in_x = timed(lambda: check(lib.wm_lfss_in_fwd(_ptr(tok), 1, _ptr(ln1w), _ptr(ln1b), 1e-5, _ptr(Win), _ptr(x), None, B, L, C, 0, st), "in"))
    t_mid = timed(lambda: check(lib.wm_lfss_mid_fwd(_ptr(y4), 4, B * D * L, _ptr(z), _ptr(tok), 1, *args_tail), "mid"))
    f0, t0 = f.clone(), tok1.clone()
    t_rz = timed(lambda: check(lib.wm_lfss_mid_rz_fwd(_ptr(y4), 4, B * D * L, _ptr(tok), 1, _ptr(ln1w), _ptr(ln1b), 1e-5, _ptr(Win), *args_tail), "rz"))
    same = torch.equal(f, f0) and torch.equal(tok1, t0)
    print(f"level {lvl} {H}x{W}: lfss_in {t_in:.3f} -> {t_in_x:.3f} ms (x half only), lfss_mid {t_mid:.3f} -> {t_rz:.3f} ms (gate recomputed), "
          f"pair {t_in + t_mid:.3f} -> {t_in_x + t_rz:.3f} ms; outputs bit-identical: {same}")
