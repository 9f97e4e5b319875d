#!/usr/bin/env python3
"""bf16 planes + multi-stream order at UHD (tools/debug_bf16_determinism.py): a device synchronisation in FRONT of one operator
class at a time - which one has to wait for the side streams for the result to become the single-stream one?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wave_mamba_amd as wm
import bench
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = wm.WaveMamba(**bench.SHIPPED).eval().to(dev)
unet = net.restoration_network
g = torch.Generator().manual_seed(1234)
x = torch.rand(1, 3, 2176, 3840, generator=g).to(dev)
# force the side streams in bf16 mode (UNet.forward takes the single-stream order for bf16 planes since the end of round 4)
real_get = wm.ops.get_plane_dtype
names = ["lfss_block_forward", "dwt_init", "iwt_init_pair", "conv2d", "dwconv3x3", "skff", "gram", "conv2d_gated", "conv2d_ln",
         "layernorm2d", "patchify_conv", "attn_fold", "match_index"]
orig = {n: getattr(wm.ops, n) for n in names if hasattr(wm.ops, n)}
def wrap(n):
    f = orig[n]
    def w(*a, **k):
        torch.cuda.synchronize()
        return f(*a, **k)
    return w
with torch.no_grad():
    wm.ops.set_plane_dtype(torch.bfloat16)
    unet.two_streams = False
    base = unet(x); torch.cuda.synchronize()
    unet.two_streams = True
    wm.ops.get_plane_dtype = lambda: torch.float32          # (only UNet.forward's stream-order switch reads it)
    for n in [None] + list(orig):
        for m in orig: setattr(wm.ops, m, orig[m])
        if n: setattr(wm.ops, n, wrap(n))
        d = []
        for _ in range(3):
            o = unet(x); torch.cuda.synchronize(); d.append(float((o - base).abs().max()))
        print(f"sync in front of every {n or '(nothing)'}: max |diff| vs single-stream {['%.2e' % v for v in d]}", flush=True)
    wm.ops.get_plane_dtype = real_get
    wm.ops.set_plane_dtype(torch.float32)
