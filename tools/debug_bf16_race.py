#!/usr/bin/env python3
"""bf16 planes + multi-stream order at UHD is not bit-reproducible (tools/debug_bf16_determinism.py): which overlap matters?
Device synchronisations inserted after selected groups of the forward."""
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
H, W = (int(v) for v in os.environ.get("HW", "2176x3840").split("x"))
x = torch.rand(1, 3, H, W, generator=g).to(dev)
SYNC = set()
def hook(name):
    def h(mod, inp, out):
        if name in SYNC: torch.cuda.synchronize()
    return h
names = ["down_group1", "down_group2", "down_group3", "up_group3", "up_group2", "up_group1"]
for n in names: getattr(unet, n).register_forward_hook(hook(n))
with torch.no_grad():
    wm.ops.set_plane_dtype(torch.bfloat16)
    unet.two_streams = False
    base = unet(x); torch.cuda.synchronize()
    unet.two_streams = True
    for cfg in ([], names, ["down_group1"], ["down_group2"], ["down_group3"], ["down_group1", "down_group2", "down_group3"], ["up_group3", "up_group2", "up_group1"]):
        SYNC.clear(); SYNC.update(cfg)
        d = []
        for _ in range(4):
            o = unet(x); torch.cuda.synchronize(); d.append(float((o - base).abs().max()))
        print(f"{H}x{W} sync after {cfg or 'nothing'}: max |diff| vs single-stream {['%.2e' % v for v in d]}", flush=True)
    wm.ops.set_plane_dtype(torch.float32)
