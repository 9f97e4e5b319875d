#!/usr/bin/env python3
"""Where does the fused-core backward differ from autograd through the unfused path?  Error of dx by direction, batch item,
channel, position within the 256-step block and step within the 16-step chunk."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import wave_mamba_amd as wm
from test_gpu_parity import random_core_case
dev = "cuda:0"
B, D, H, W, N, R = [int(v) for v in (sys.argv[1:7] if len(sys.argv) > 6 else (8, 64, 256, 256, 16, 2))]
x, Wx, Wdt, bias, A_logs, Ds = [t.to(dev).requires_grad_(True) for t in random_core_case(B, D, H, W, N, R, seed=H + W + B)]
L = H * W
params = [x, Wx, Wdt, bias, A_logs, Ds]
gg = torch.Generator(device=dev).manual_seed(7)
dy1 = torch.randn(B, D, L, device=dev, generator=gg)

def unfused():
    xs = torch.stack([x.view(B, -1, L), x.transpose(2, 3).contiguous().view(B, -1, L)], dim=1).view(B, 2, -1, L)
    xs = torch.cat([xs, torch.flip(xs, dims=[-1])], dim=1)
    x_dbl = torch.einsum("b k d l, k c d -> b k c l", xs, Wx)
    dts, Bs, Cs = torch.split(x_dbl, [R, N, N], dim=2)
    dts = torch.einsum("b k r l, k d r -> b k d l", dts, Wdt)
    out = wm.ops.selective_scan_fn(xs.reshape(B, -1, L), dts.reshape(B, -1, L), -torch.exp(A_logs), Bs.contiguous(),
                                   Cs.contiguous(), Ds, None, bias.reshape(-1), True).view(B, 4, -1, L)
    inv = torch.flip(out[:, 2:4], dims=[-1]).view(B, 2, -1, L)
    wh = out[:, 1].view(B, -1, W, H).transpose(2, 3).contiguous().view(B, -1, L)
    invwh = inv[:, 1].view(B, -1, W, H).transpose(2, 3).contiguous().view(B, -1, L)
    return out[:, 0], inv[:, 0], wh, invwh

names = ("row fwd", "row rev", "col fwd", "col rev")
z = torch.zeros_like(dy1)
for k in range(4):
    dys = [dy1 if i == k else z for i in range(4)]
    ref = torch.autograd.grad(unfused(), params, dys)
    for rep in range(2):
        got = torch.autograd.grad(wm.ops.ss2d_core(*params), params, dys)
        e = (got[0] - ref[0]).abs()
        scale = float(ref[0].abs().max())
        print(f"{names[k]} run {rep}: dx rel max {float(e.max()) / scale:.3e}, rel l2 {float((got[0] - ref[0]).norm() / ref[0].norm()):.3e}; "
              + " ".join(f"{nm} {float((g - r).norm() / r.norm()):.1e}" for nm, g, r in zip(("dWx", "dWdt", "db", "dA", "dD"), got[1:], ref[1:])))
        bad = e > 1e-3 * scale
        if bad.any():
            idx = bad.nonzero()
            bb, cc, hh, ww = idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]
            pos = (hh * W + ww) if k % 2 == 0 else (ww * H + hh)          # position in the layout the direction scans
            print("   channels", sorted(set(cc.tolist())), "first bad (b, c, pos, got, ref):",
                  [(int(bb[i]), int(cc[i]), int(pos[i]), round(float(got[0][bb[i], cc[i], hh[i], ww[i]]), 4), round(float(ref[0][bb[i], cc[i], hh[i], ww[i]]), 4)) for i in range(min(6, len(bb)))])
            # errors of one (b, block of 16 positions): which channels / steps
            i0 = 0
            sel = (bb == bb[i0]) & ((pos // 16) == (pos[i0] // 16))
            print("   in that chunk: (channel, step) of the bad elements:", sorted(set(zip(cc[sel].tolist(), (pos[sel] % 16).tolist())))[:40])
            print(f"   {int(bad.sum())} bad elements; batch items {sorted(set(bb.tolist()))[:8]}; channels {len(set(cc.tolist()))} distinct "
                  f"(min {int(cc.min())} max {int(cc.max())}); pos % 16 histogram {torch.bincount(pos % 16, minlength=16).tolist()}; "
                  f"(pos // 16) % 16 histogram {torch.bincount((pos // 16) % 16, minlength=16).tolist()}; blocks {len(set((pos // 256).tolist()))} of {L // 256}")
