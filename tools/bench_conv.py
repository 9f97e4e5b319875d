"""Dense 3x3 / 1x1 convolution (wm_conv2d_fwd; KS=1|3 in the environment) against MIOpen / hipBLASLt (F.conv2d fp32) at the UHD pyramid shapes:
accuracy of both against an fp64 convolution, and time per call.  GPU only."""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wave_mamba_amd as wm  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
KS = int(os.environ.get("KS", "3"))
shapes = [  # (Ca, Cb, Cout, H, W, bias)
    (64, 0, 64, 1088, 1920, False), (64, 0, 32, 1088, 1920, False), (32, 32, 32, 1088, 1920, True),
    (32, 0, 96, 1088, 1920, True), (64, 0, 64, 544, 960, False), (64, 0, 64, 272, 480, False),
    (64, 0, 32, 272, 480, False), (3, 0, 32, 2176, 3840, True), (32, 0, 3, 2176, 3840, True),
]
if len(sys.argv) > 1 and sys.argv[1] == "one":
    shapes = shapes[:1]
if len(sys.argv) > 1 and sys.argv[1] == "l1":
    shapes = shapes[:3] + shapes[7:]
if len(sys.argv) > 1 and sys.argv[1] == "small":
    shapes = [(64, 0, 64, 70, 50, True), (16, 8, 40, 33, 65, True), (3, 0, 32, 40, 64, True), (32, 0, 3, 64, 96, False)]


def timeit(fn, n=int(os.environ.get("WM_BENCH_N", "5"))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for ca, cb, co, H, W, hb in shapes:
    xa = torch.randn(1, ca, H, W, device=dev, generator=g)
    xb = torch.randn(1, cb, H, W, device=dev, generator=g) if cb else None
    w = torch.randn(co, ca + cb, KS, KS, device=dev, generator=g) / (KS * (ca + cb) ** 0.5)
    b = torch.randn(co, device=dev, generator=g) if hb else None
    xin = xa if xb is None else torch.cat([xa, xb], 1)
    ref32 = F.conv2d(xin, w, b, padding=KS // 2)
    out = wm.ops.conv2d(xa, w, b, xb)
    # fp64 reference on a crop (whole image for the small ones)
    hh, ww = min(H, 256), min(W, 256)
    ref64 = F.conv2d(xin[:, :, :hh + 1, :ww + 1].double(), w.double(), None if b is None else b.double(), padding=KS // 2)[:, :, :hh, :ww]
    def rel(a):
        a = a[:, :, :hh, :ww].double()
        return float((a - ref64).norm() / ref64.norm()), float((a - ref64).abs().max() / ref64.abs().max())
    t_wm = timeit(lambda: wm.ops.conv2d(xa, w, b, xb))
    t_mi = float("nan") if os.environ.get("WM_NO_MIOPEN") else \
        timeit(lambda: F.conv2d(xin if xb is None else torch.cat([xa, xb], 1), w, b, padding=KS // 2))
    flops = 2.0 * KS * KS * (ca + cb) * co * H * W
    byts = 4.0 * (ca + cb + co) * H * W
    print(f"Cin {ca}+{cb} Cout {co} {H}x{W}: wm {t_wm:.3f} ms ({flops / t_wm / 1e9:.0f} TFLOP/s fp32-equiv, "
          f"{byts / t_wm / 1e6:.0f} GB/s)  miopen {t_mi:.3f} ms | rel_l2/max vs fp64: wm {rel(out)[0]:.2e}/{rel(out)[1]:.2e} "
          f"miopen {rel(ref32)[0]:.2e}/{rel(ref32)[1]:.2e} | wm vs miopen max {float((out - ref32).abs().max()):.2e}", flush=True)
