#!/usr/bin/env python3
"""Decode WHICH term of the depth-wise 3x3 (bf16 planes, no activation) goes wrong under the conv3x3 aggressor
(tools/repro_pk_lanes.py): one-hot tap weights / bias only, inputs that encode their position."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wave_mamba_amd as wm

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(7)
N = int(os.environ.get("N", "400"))
H, W = 272, 480
xa = torch.randn(1, 64, 544, 960, generator=g).to(dev)
w3 = (torch.randn(64, 64, 3, 3, generator=g) / 24).to(dev)
side = torch.cuda.Stream(device=dev)
# x[c][r][w] = small integers exactly representable in bf16: row code + column code
rr = torch.arange(H).view(1, 1, H, 1).float()
cc = torch.arange(W).view(1, 1, 1, W).float()
x = ((rr % 8) * 16 + (cc % 16) + 1).expand(1, 64, H, W).contiguous().to(dev).bfloat16()       # 1 .. 128, exact in bf16
with torch.no_grad():
    for name in [f"tap{i}" for i in range(9)] + ["bias", "all"]:
        wgt = torch.zeros(64, 1, 3, 3)
        b = torch.zeros(64)
        if name.startswith("tap"):
            wgt.view(64, 9)[:, int(name[3:])] = 1.0
        elif name == "bias":
            b[:] = 3.0
        else:
            wgt[:] = torch.tensor([1., 2., 4., 8., 16., 32., 64., 128., 256.]).view(3, 3) / 256.0; b[:] = 0.0
        wgt, b = wgt.to(dev), b.to(dev)
        ref = wm.ops.dwconv3x3(x, wgt, b, "none"); torch.cuda.synchronize()
        bad = 0; shown = 0
        for i in range(N):
            with torch.cuda.stream(side):
                keep = wm.ops.conv2d(xa, w3)
            o = wm.ops.dwconv3x3(x, wgt, b, "none")
            torch.cuda.synchronize()
            ne = o.float() != ref.float()
            if bool(ne.any()):
                bad += 1
                if shown < 2:
                    shown += 1
                    idx = ne.nonzero()
                    print(f"  {name}: {int(ne.sum())} elements; lanes {sorted(set((int(c) // 4) % 64 for c in idx[:, 3].tolist()))[:4]}.. "
                          f"quad elements {sorted(set(int(c) % 4 for c in idx[:, 3].tolist()))} rows {torch.unique(idx[:, 2]).tolist()[:6]} ch {torch.unique(idx[:, 1]).tolist()[:6]}")
                    for j in idx[:5].tolist():
                        print(f"      [ch {j[1]}, row {j[2]}, col {j[3]}] got {float(o[tuple(j)]):g} expected {float(ref[tuple(j)]):g}")
        print(f"{name}: {bad} of {N} launches differ", flush=True)
