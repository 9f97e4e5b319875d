#!/usr/bin/env python3
"""SS2D's prologue at the three UHD pyramid levels: wm_lfss_in_conv_fwd (one kernel) against wm_lfss_in_fwd + wm_dwconv3x3_fwd."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wave_mamba_amd as wm
from wave_mamba_amd.archs import wavemamba_arch as arch
dev = "cuda:0"
torch.manual_seed(0)
blk = arch.LFSSBlock(32, expand=2.0).to(dev).eval()
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (H, W) in [(1088, 1920), (544, 960), (272, 480)]:
    tok = torch.randn(1, H * W, 32, device=dev)
    a = t(lambda: wm.ops.lfss_prologue(tok, (H, W), blk, fused=True))
    b = t(lambda: wm.ops.lfss_prologue(tok, (H, W), blk, fused=False))
    pos = H * W
    print(f"{H} x {W}: one kernel {a:.3f} ms ({640 * pos / a / 1e6:.0f} GB/s on 640 B / position) | two kernels {b:.3f} ms ({1152 * pos / b / 1e6:.0f} GB/s on 1152 B / position)")
