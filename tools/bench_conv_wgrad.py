#!/usr/bin/env python3
"""Weight gradient of the dense convolutions at the BASELINE config-3 training sizes: wm_conv2d_wgrad against ATen's
convolution_backward(output_mask = weight only), ms per call (HIP events, 20 calls after 5 warm-ups)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wave_mamba_amd as wm
dev = torch.device("cuda", 0)
def timeit(f, n=20):
    for _ in range(5): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
shapes = [(8, 64, 64, 256, 256, 3), (8, 64, 32, 256, 256, 3), (8, 64, 64, 128, 128, 3), (8, 64, 64, 64, 64, 3), (8, 3, 32, 512, 512, 3),
          (8, 32, 96, 256, 256, 3), (8, 32, 32, 256, 256, 1), (8, 32, 64, 256, 256, 1), (8, 64, 64, 256, 256, 1), (8, 32, 96, 256, 256, 1),
          (8, 32, 32, 64, 64, 1)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for (B, Cin, Cout, H, W, ks) in shapes:
    x = torch.randn(B, Cin, H, W, device=dev); gy = torch.randn(B, Cout, H, W, device=dev)
    w = torch.randn(Cout, Cin, ks, ks, device=dev)
    t_hip = timeit(lambda: wm.ops.conv2d_wgrad(gy, x, ks))
    t_at = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [ks // 2] * 2, [1, 1], False, [0, 0], 1, [False, True, False]))
    flops = 2.0 * B * H * W * Cin * Cout * ks * ks
    print(f"B{B} {Cin:3d}->{Cout:3d} {H}x{W} ks{ks}: HIP {t_hip:7.3f} ms ({flops / t_hip / 1e9:7.1f} TFLOP/s useful)   ATen {t_at:7.3f} ms")
