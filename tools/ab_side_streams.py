#!/usr/bin/env python3
"""A/B of the inference forward's side-stream arrangement on one box: which levels' high-frequency branches leave the main stream,
on how many streams, at which stream priority.  ms per padded UHD forward (20 timed, 5 warm-up), two rounds."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wave_mamba_amd as wm
import bench
from wave_mamba_amd.archs import wavemamba_arch as arch
dev = torch.device("cuda", 0)
net = bench.build_model(dev)
unet = net.restoration_network
x = bench.pad_to(torch.rand(1, 3, 2160, 3840, generator=torch.Generator().manual_seed(1234))).to(dev)
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("stream priority range (least, greatest):", lo, hi)
S = {p: [torch.cuda.Stream(device=dev, priority=p) for _ in range(3)] for p in sorted({0, lo, hi})}
VARIANTS = {
    "three side streams, default priority (shipped)": lambda: tuple(S[0]),
    f"three side streams, priority {lo} (least)": lambda: tuple(S[lo]),
    f"three side streams, priority {hi} (greatest)": lambda: tuple(S[hi]),
    "one side stream for all levels": lambda: (S[0][0],) * 3,
    "level 1 only": lambda: (S[0][0], None, None),
    "levels 2 + 3 only": lambda: (None, S[0][1], S[0][2]),
    "levels 1 + 2 only": lambda: (S[0][0], S[0][1], None),
    "single stream": lambda: (None, None, None),
}
MAIN_HI = torch.cuda.Stream(device=dev, priority=hi)
VARIANTS[f"three side streams (default priority), the forward itself on a stream of priority {hi}"] = lambda: tuple(S[0])
VARIANTS[f"single stream, of priority {hi}"] = lambda: (None, None, None)
orig = arch._side_streams
ref = None
for rnd in range(2):
    for name, mk in VARIANTS.items():
        arch._side_streams = lambda x_, n, mk=mk: mk()[:n]
        main = MAIN_HI if "forward itself" in name or name.startswith("single stream, of") else torch.cuda.current_stream(dev)
        main.wait_stream(torch.cuda.current_stream(dev))
        with torch.no_grad(), torch.cuda.stream(main):
            for _ in range(5):
                y = unet(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                y = unet(x)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 20
        if ref is None:
            ref = y.clone()
        print(f"round {rnd}: {dt * 1e3:7.3f} ms  {name}   (bit-equal to the first variant: {torch.equal(y, ref)})", flush=True)
arch._side_streams = orig
