#!/usr/bin/env python3
"""Diagnosis of the multi-stream mismatch (profiles/r04/multi_stream_patchify_mismatch.txt): the fused patch embeddings forced back onto
the side streams (the order that failed), under variations that separate the hypotheses - which level's embedding matters, whether a HOST
wait for the main stream's work before the side streams start removes it (then the device-side wait does not order what it should),
a delay kernel in front, and a non-default main stream."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wave_mamba_amd as wm
from wave_mamba_amd.archs import wavemamba_arch as arch
DEV = "cuda:0"
gen = lambda s: torch.Generator().manual_seed(s)
net = wm.WaveMamba(in_chn=3, wf=32, n_l_blocks=[1, 2, 4], n_h_blocks=[1, 1, 2], ffn_scale=2.0).eval().to(DEV)
unet = net.restoration_network
xs = [torch.rand(1, 3, 264, 392, generator=gen(41)).to(DEV), torch.rand(2, 3, 136, 200, generator=gen(42)).to(DEV)]
REPS = int(os.environ.get("REPS", "60"))
orig_sup = wm.ops.patchify_conv_supported
hidden = lambda *a: False
MODE = {"levels": (2, 4, 8), "pre": None}


def ps_side(ps, img):
    r, conv = ps[0].downscale_factor, ps[1]
    if r not in MODE["levels"]:                               # this level: the two-module form (as before the kernel existed)
        return arch._conv(conv, ps[0](img))
    if MODE["pre"] == "host_wait":
        torch.cuda.synchronize()                              # everything issued so far is DONE before this side stream's kernel
    elif MODE["pre"] == "sleep":
        torch.cuda._sleep(200000)
    wm.ops.patchify_conv_supported = orig_sup
    try:
        return wm.ops.patchify_conv(img, conv.weight, conv.bias, r)
    finally:
        wm.ops.patchify_conv_supported = hidden


def run(tag, stream=None):
    bad = 0
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        for rep in range(REPS):
            with torch.no_grad():
                unet.two_streams = False
                refs = [unet(x) for x in xs]
                torch.cuda.synchronize()
                unet.two_streams = True
                for _ in range(3):
                    outs = [unet(x) for x in xs]
            torch.cuda.synchronize()
            bad += 0 if torch.equal(refs[0], outs[0]) else 1
    print(f"{tag}: mismatching x0 forwards {bad} of {REPS}", flush=True)


arch._ps_conv = ps_side
wm.ops.patchify_conv_supported = hidden          # -> side-stream branch of UNet.forward
run("embeddings of all three levels on the side streams")
for lv in ((2,), (4,), (8,)):
    MODE["levels"] = lv
    run(f"only the r = {lv[0]} embedding fused (the others: two-module form), side streams")
MODE["levels"] = (2, 4, 8)
MODE["pre"] = "host_wait"; run("all three, host synchronisation before each side-stream embedding")
MODE["pre"] = "sleep"; run("all three, a 200k-cycle delay kernel on the side stream in front of each")
MODE["pre"] = None
other = torch.cuda.Stream(DEV)
other.wait_stream(torch.cuda.current_stream(DEV))
run("all three, the forwards issued on a NON-default main stream", stream=other)
torch.cuda.current_stream(DEV).wait_stream(other)
run("all three on the side streams (again)")
