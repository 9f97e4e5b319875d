#!/usr/bin/env python3
"""PCIe-inclusive rate of the inference counterpart: host uint8 UHD images in, host uint8 images out
(wave_mamba_amd.inference.UInt8Pipeline: pinned double-buffered H2D / D2H under the forward).  GPU only."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wave_mamba_amd as wm
from wave_mamba_amd import inference
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = wm.WaveMamba(in_chn=3, wf=32, n_l_blocks=[1, 2, 4], n_h_blocks=[1, 1, 2], ffn_scale=2.0).eval().to(dev)
rng = np.random.default_rng(0)
imgs = [rng.integers(0, 256, size=(2160, 3840, 3), dtype=np.uint8) for _ in range(4)]
pipe = inference.UInt8Pipeline(net, dev)
list(pipe.run(imgs[:3]))                      # warm-up
n = 16
t0 = time.perf_counter()
cnt = sum(1 for _ in pipe.run(imgs[i % 4] for i in range(n)))
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"{cnt} UHD uint8 images host -> host in {el:.3f} s: {cnt / el:.2f} images/s ({1e3 * el / cnt:.1f} ms per image, "
      f"24.9 MB up + 24.9 MB down each)")
