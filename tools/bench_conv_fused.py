"""wm_conv2d_fwd with the HFE fusions (gathered second operand, sigmoid gate, residual) against the plain launch."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wave_mamba_amd as wm
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
H, W = 1088, 1920
x = torch.randn(1, 32, H, W, device=dev, generator=g); p = torch.randn(1, 32, H, W, device=dev, generator=g)
x64 = torch.randn(1, 64, H, W, device=dev, generator=g)
idx = torch.randint(0, 32, (1, 32), device=dev, generator=g, dtype=torch.int64).int()
gate = torch.randn(1, 64, H, W, device=dev, generator=g); res = torch.randn(1, 64, H, W, device=dev, generator=g)
def timeit(fn, n=int(os.environ.get("WM_BENCH_N", "5"))):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for ks in (3, 1):
    w = torch.randn(64, 64, ks, ks, device=dev, generator=g) / (8 * ks)
    print(f"ks={ks} plain {timeit(lambda: wm.ops.conv2d(x64, w)):.3f}  cat {timeit(lambda: wm.ops.conv2d(x, w, None, p)):.3f}  "
          f"gather {timeit(lambda: wm.ops.conv2d(x, w, None, p, idx)):.3f}  gather+gate {timeit(lambda: wm.ops.conv2d(x, w, None, p, idx, gate)):.3f}  "
          f"plain+res {timeit(lambda: wm.ops.conv2d(x64, w, None, None, None, None, res)):.3f}  plain+gate {timeit(lambda: wm.ops.conv2d(x64, w, None, None, None, gate)):.3f} ms")
w3 = torch.randn(64, 64, 3, 3, device=dev, generator=g) / 24
w1 = torch.randn(64, 64, 1, 1, device=dev, generator=g) / 8
b1 = torch.randn(64, device=dev, generator=g)
print(f"k3 * sigmoid(k2) in one kernel (PAConv): plain {timeit(lambda: wm.ops.conv2d_gated(x64, w3, w1, b1)):.3f}  "
      f"gather {timeit(lambda: wm.ops.conv2d_gated(x, w3, w1, b1, p, idx)):.3f} ms")
for (hh, ww) in ((544, 960), (272, 480)):
    xs = torch.randn(1, 64, hh, ww, device=dev, generator=g)
    print(f"{hh}x{ww}: 64->64 {timeit(lambda: wm.ops.conv2d(xs, w3)):.4f}  64->32 {timeit(lambda: wm.ops.conv2d(xs, w3[:32])):.4f}  "
          f"gated {timeit(lambda: wm.ops.conv2d_gated(xs, w3, w1, b1)):.4f} ms")
