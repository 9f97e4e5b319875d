"""Phase time stamps of workgroup 0 of the wave-specialised 3x3 convolution (library built with -DWM_CV_STAMP=1):
consumer wave 0 (mma / barrier wait / epilogue) and producer wave 4 (issue / stage / vmcnt wait / barrier wait) per chunk step."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wave_mamba_amd as wm  # noqa: E402
from wave_mamba_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
ca, co, H, W = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (64, 64, 1088, 1920)))
x = torch.randn(1, ca, H, W, device=dev)
w = torch.randn(co, ca, 3, 3, device=dev) / (3 * ca ** 0.5)
for _ in range(3):
    wm.ops.conv2d(x, w)
torch.cuda.synchronize()
lib = _lib.load() if hasattr(_lib, "load") else _lib.lib
buf = (ctypes.c_ulonglong * (2 * 128 * 8))()
lib.wm_debug_conv_stamps.restype = ctypes.c_int
rc = lib.wm_debug_conv_stamps(buf)
assert rc == 0, rc
st = np.frombuffer(buf, dtype=np.uint64).reshape(2, 128, 8).astype(np.int64)
t0 = st[0, 0, 0]
print("consumer wave 0: step | mma | barrier wait | epilogue | step total")
for i in range(0, 16):
    c = st[0, i]
    nxt = st[0, i + 1, 0]
    print(f"  {i:3d} start {c[0] - t0:8d}  mma {c[1] - c[0]:6d}  wait {c[2] - c[1]:6d}  epi {c[3] - c[2]:6d}  total {nxt - c[0]:6d}")
print("producer wave 4: step | issue (DMA + loads) | stage | vmcnt wait | barrier wait")
for i in range(0, 16):
    p = st[1, i]
    print(f"  {i:3d} start {p[0] - t0:8d}  issue {p[1] - p[0]:6d}  stage {p[2] - p[1]:6d}  vmcnt {p[3] - p[2]:6d}  barrier {p[4] - p[3]:6d}")
