#!/usr/bin/env python3
"""ps_down1..3 on the UHD image (1 x 3 x 2176 x 3840): nn.PixelUnshuffle(r) + the 1x1 matrix-core convolution (two launches, the
unshuffled tensor materialised by a strided copy) against wm_patchify_conv_fwd (one r x r / stride-r kernel).  ms per call."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wave_mamba_amd as wm
dev = "cuda:0"
g = torch.Generator(device=dev); g.manual_seed(0)
img = torch.rand(1, 3, 2176, 3840, device=dev, generator=g)


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for r in (2, 4, 8):
    w = torch.randn(32, 3 * r * r, 1, 1, device=dev, generator=g) / (3 * r * r) ** 0.5
    b = torch.randn(32, device=dev, generator=g)
    t_copy = timed(lambda: F.pixel_unshuffle(img, r))
    t_two = timed(lambda: wm.ops.conv2d(F.pixel_unshuffle(img, r), w, b))
    t_one = timed(lambda: wm.ops.patchify_conv(img, w, b, r))
    ref = F.conv2d(F.pixel_unshuffle(img.double(), r), w.double(), b.double())
    e_two = float((wm.ops.conv2d(F.pixel_unshuffle(img, r), w, b).double() - ref).norm() / ref.norm())
    e_one = float((wm.ops.patchify_conv(img, w, b, r).double() - ref).norm() / ref.norm())
    gb = (img.numel() + ref.numel()) * 4 / 1e9
    print(f"r = {r}: unshuffle copy {t_copy:.3f} ms, copy + 1x1 conv {t_two:.3f} ms (rel err {e_two:.1e}) -> one kernel {t_one:.3f} ms "
          f"(rel err {e_one:.1e}; {gb:.3f} GB moved = {gb / t_one:.2f} TB/s)")
