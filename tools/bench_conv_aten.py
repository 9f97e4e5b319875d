#!/usr/bin/env python3
"""ATen / MIOpen fp32 3x3 convolution at BASELINE config 3's sizes: the forward solver (Winograd f2x3, 8.6 ms of a 70-ms step for
79 calls) against the SAME convolution evaluated through the input-gradient entry point (aten.convolution_backward with
grad_output := x and weight := flip(W)^T; MIOpen picks its f3x2 kernel there: 2.8 ms for 56 calls in the same step)."""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = "cuda:0"
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (B, Ci, Co, H, W) in [(8, 64, 64, 256, 256), (8, 64, 32, 256, 256), (8, 32, 96, 256, 256), (8, 3, 32, 512, 512), (8, 64, 64, 128, 128), (8, 64, 64, 64, 64)]:
    x = torch.randn(B, Ci, H, W, device=dev); w = torch.randn(Co, Ci, 3, 3, device=dev) * 0.05
    ref = F.conv2d(x, w, None, padding=1)
    wb = w.flip(2, 3).transpose(0, 1).contiguous()            # (Ci, Co, 3, 3): weight of the convolution whose input gradient is conv(x, w)
    dummy = torch.empty(B, Co, H, W, device=dev)
    via = lambda: torch.ops.aten.convolution_backward(x, dummy, wb, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])[0]
    err = float((via() - ref).abs().max() / ref.abs().max())
    t_f = t(lambda: F.conv2d(x, w, None, padding=1)); t_b = t(via)
    gy = torch.randn_like(ref)
    t_bd = t(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])[0])
    print(f"{B}x{Ci}->{Co} {H}x{W}: forward {t_f:.3f} ms; forward via the input-gradient entry {t_b:.3f} ms (max dev {err:.1e}); real input gradient {t_bd:.3f} ms")
