#!/usr/bin/env python3
"""Is the bf16-plane UHD forward bit-reproducible?  N synchronised forwards, single-stream and multi-stream order."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wave_mamba_amd as wm
import bench
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = wm.WaveMamba(**bench.SHIPPED).eval().to(dev)
g = torch.Generator().manual_seed(1234)
x = torch.rand(1, 3, 2176, 3840, generator=g).to(dev)
N = int(os.environ.get("REPS", "8"))
print("WM_LFSS_OUT_ROWS =", os.environ.get("WM_LFSS_OUT_ROWS", "-1"))
with torch.no_grad():
    for dt in (torch.float32, torch.bfloat16):
        wm.ops.set_plane_dtype(dt)
        for two in (False, True):
            net.restoration_network.two_streams = two
            outs = []
            for _ in range(N):
                outs.append(net.restoration_network(x)); torch.cuda.synchronize()
            diff = [float((o - outs[0]).abs().max()) for o in outs[1:]]
            print(f"planes {dt}, two_streams {two}: max |diff| to the first forward: {['%.2e' % d for d in diff]}", flush=True)
            if two is False: base = outs[0]
            else: print(f"   multi-stream vs single-stream: {float((outs[0] - base).abs().max()):.3e}")
    wm.ops.set_plane_dtype(torch.float32)
