#!/usr/bin/env python3
"""tools/ubench_dwconv_variants.hip against the library's 3x3 convolution on a second stream: launches (of N) whose output differs
from the launch that ran alone, per source variant."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wave_mamba_amd as wm
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "build", "ubench_dwconv_variants.so"))
P, I = ctypes.c_void_p, ctypes.c_int
lib.dwv_launch.argtypes = [I, P, P, P, P, I, I, I, ctypes.c_longlong, P]
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(7)
N = int(os.environ.get("N", "300"))
H, W = 272, 480
x = torch.randn(1, 64, H, W, generator=g).to(dev)
xb = x.bfloat16()
wgt = (torch.randn(64, 1, 3, 3, generator=g) / 3).to(dev).contiguous()
b = torch.randn(64, generator=g).to(dev)
xa = torch.randn(1, 64, 544, 960, generator=g).to(dev)
w3 = (torch.randn(64, 64, 3, 3, generator=g) / 24).to(dev)
side = torch.cuda.Stream(device=dev)
NAMES = {0: "as shipped (bf16 -> bf16)", 1: "tap weights in VGPRs", 2: "no wave shuffles", 3: "loads waited for at once",
         5: "bf16 -> fp32 (no pack)", 6: "fp32 -> bf16", 7: "SLP defeated (no packed fp32)", 8: "halo from a dword load (no ushort)"}


def run(var):
    xin = x if var == 6 else xb
    y = torch.empty((1, 64, H, W), dtype=torch.float32 if var == 5 else torch.bfloat16, device=dev)
    rc = lib.dwv_launch(var, xin.data_ptr(), wgt.data_ptr(), b.data_ptr(), y.data_ptr(), 64, H, W, 64, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    return y


wm.ops.conv2d_select(wm.ops.CONV3X3_FIRST_GEN)
with torch.no_grad():
    for var, name in NAMES.items():
        ref = run(var); torch.cuda.synchronize()
        cnts, keep = [], []
        for i in range(N):
            with torch.cuda.stream(side):
                keep.append(wm.ops.conv2d(xa, w3))
                if len(keep) > 6:
                    keep.pop(0)
            o = run(var)
            cnts.append((o.float() != ref.float()).sum())
        torch.cuda.synchronize()
        bad = [int(c) for c in cnts]
        print(f"variant {var} {name:32s}: {sum(1 for c in bad if c):4d} of {N} launches differ", flush=True)
