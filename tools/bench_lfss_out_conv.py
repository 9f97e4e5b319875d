#!/usr/bin/env python3
"""wm_lfss_out_conv_fwd (the ffn's depth-wise 3x3 + gelu gate + conv3 + scaled skip, one kernel) at the three UHD levels, ms per call
(HIP events) and a checksum of the output.  Levels 1 and 2 (W % 64 == 0) run the accumulating row-window form (R = 4, round 6), level 3
the one-row banded form; profiles/r06/lfss_out_conv_forms.txt holds the forms measured against each other."""
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
cw, cb, W3, b3, sk2 = rn(D, 1, 3, 3) / 3, rn(D) * 0.1, rn(C, C) / 6, rn(C) * 0.1, rn(C) * 0.1 + 1
for lvl in (1, 2, 3):
    H, W = 2176 >> lvl, 3840 >> lvl
    L, B = H * W, 1
    f, tok1, out = rn(B, D, H, W), rn(B, L, C), torch.empty(B, L, C, device=dev)
    st = _stream()
    fn = lambda: check(lib.wm_lfss_out_conv_fwd(_ptr(f), _ptr(cw), _ptr(cb), _ptr(tok1), _ptr(W3), _ptr(b3), _ptr(sk2), _ptr(out), 0,
                                                B, H, W, C, 0, st), "out_conv")
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20
    print(f"level {lvl} {H}x{W}: {t:.3f} ms = {512 * L / t / 1e9:.2f} TB/s on 512 B per position; checksum {float(out.double().sum()):.9e}")
