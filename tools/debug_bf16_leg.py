#!/usr/bin/env python3
"""Is the fp32 reference forward of bench.py's bf16 leg reproducible when the bf16 forwards are issued right behind it (no host
synchronisation)?  The UHD image, N repetitions: rel l2 of each `ref` against a synchronised fp32 forward."""
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
N = int(os.environ.get("REPS", "12"))
bad = 0
with torch.no_grad():
    clean = net.restoration_network(x); torch.cuda.synchronize()
    for rep in range(N):
        prev = wm.ops.set_plane_dtype(torch.float32)
        ref = net.restoration_network(x)
        wm.ops.set_plane_dtype(torch.bfloat16)
        for _ in range(2):
            out = net.restoration_network(x)
        torch.cuda.synchronize()
        wm.ops.set_plane_dtype(prev)
        if not torch.equal(ref, clean):
            bad += 1
            print("rep", rep, "fp32 forward followed by bf16 forwards differs from the synchronised one: rel l2", float((ref - clean).norm() / clean.norm()), flush=True)
print("fp32 forwards that changed under a following bf16 forward:", bad, "of", N)
