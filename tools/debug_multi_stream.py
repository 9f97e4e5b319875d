#!/usr/bin/env python3
"""The multi-stream forward against the single-stream forward (tests/test_gpu_parity.py::test_multi_stream_forward_is_the_single_stream_forward),
repeated, with the fused patch embedding on / off: where and how often they differ."""
import os, sys
import torch
sys.path.insert(0, os.environ.get("WM_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import wave_mamba_amd as wm
DEV = "cuda:0"
gen = lambda s: torch.Generator().manual_seed(s)
net = wm.WaveMamba(in_chn=3, wf=32, n_l_blocks=[1, 2, 4], n_h_blocks=[1, 1, 2], ffn_scale=2.0).eval().to(DEV)
unet = net.restoration_network
xs = [torch.rand(1, 3, 264, 392, generator=gen(41)).to(DEV), torch.rand(2, 3, 136, 200, generator=gen(42)).to(DEV)]
print("package:", wm.__file__)
HAVE = hasattr(wm.ops, "patchify_conv_supported")
orig = wm.ops.patchify_conv_supported if HAVE else None
REPS = int(os.environ.get("REPS", "40"))
for mode in (("fused", "unfused", "fused", "unfused") if HAVE else ("old", "old")):
    if HAVE:
        wm.ops.patchify_conv_supported = orig if mode == "fused" else (lambda *a: False)
    bad = 0
    for rep in range(REPS):
        with torch.no_grad():
            unet.two_streams = False
            refs = [unet(x) for x in xs]
            refs2 = [unet(x) for x in xs]
            unet.two_streams = True
            for _ in range(3):
                outs = [unet(x) for x in xs]
        torch.cuda.synchronize()
        for i, (r, r2, o) in enumerate(zip(refs, refs2, outs)):
            if not torch.equal(r, r2):
                print(mode, rep, i, "SINGLE-STREAM forwards differ", float((r - r2).abs().max()))
            if not torch.equal(r, o):
                bad += 1
                dmap = (r - o).abs().amax(1)      # (B, H, W)
                nz = dmap.nonzero()
                print(mode, rep, "input", i, "max diff %.3e" % float(dmap.max()), "differing pixels", nz.shape[0], "of", dmap.numel(),
                      "rows", int(nz[:, 1].min()), int(nz[:, 1].max()), "cols", int(nz[:, 2].min()), int(nz[:, 2].max()))
    print(mode, "mismatching forwards:", bad, "of", 2 * REPS)
