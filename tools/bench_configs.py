#!/usr/bin/env python3
"""Op-level numbers for BASELINE configs 4 and 5 (parity-test cases; reported in DESIGN.md, not bench lines).
   config 4: UHDLOL4K 3-level Haar DWT + IWT on 4x32x2160x4096 fp32 (11.89 GB algorithmic each way)
   config 5: selective scan with d_state 32 on a 2048x2048 map (L = 4,194,304, KD = 256): 4096 B/position"""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wave_mamba_amd as wm
dev = "cuda:0"

def timed(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

out = {}
x = torch.randn(4, 32, 2160, 4096, device=dev)
def dwt3():
    cur, pyr = x, []
    for _ in range(3):
        ll, hl, lh, hh = wm.ops.dwt_init(cur); pyr.append((hl, lh, hh)); cur = ll
    return cur, pyr
ms = timed(lambda: dwt3())
gb = 2 * 4 * (4 * 32 * 2160 * 4096) * (1 + 1 / 4 + 1 / 16) / 1e9
out["config4_dwt3"] = {"ms": ms, "algorithmic_GB": gb, "GBps": gb / ms * 1e3, "frac_of_8TBps": gb / ms * 1e3 / 8000}
cur, pyr = dwt3()
def iwt3():
    c = cur
    for hl, lh, hh in reversed(pyr):
        c = wm.ops.iwt_init_pair(c, torch.cat([hl, lh, hh], 1)) if False else wm.ops.iwt_init(torch.cat([c, hl, lh, hh], 1))
    return c
cats = [torch.cat([hl, lh, hh], 1) for hl, lh, hh in pyr]
def iwt3_pair():
    c = cur
    for h3 in reversed(cats):
        c = wm.ops.iwt_init_pair(c, h3)
    return c
ms = timed(iwt3_pair)
out["config4_iwt3"] = {"ms": ms, "algorithmic_GB": gb, "GBps": gb / ms * 1e3, "frac_of_8TBps": gb / ms * 1e3 / 8000}
del x, cur, pyr, cats
torch.cuda.empty_cache()

L, dim, N, G = 2048 * 2048, 256, 32, 4
g = torch.Generator(device=dev).manual_seed(1)
u = torch.randn(1, dim, L, device=dev, generator=g); dl = 0.5 * torch.randn(1, dim, L, device=dev, generator=g)
A = -torch.arange(1, N + 1, device=dev, dtype=torch.float32).repeat(dim, 1)
Bm = torch.randn(1, G, N, L, device=dev, generator=g); Cm = torch.randn(1, G, N, L, device=dev, generator=g)
D = torch.ones(dim, device=dev); bias = torch.full((dim,), -4.0, device=dev)
ms = timed(lambda: wm.ops.selective_scan_fn(u, dl, A, Bm, Cm, D, None, bias, True), 3)
gb = 4096 * L / 1e9
out["config5_selscan_N32"] = {"ms": ms, "algorithmic_GB": gb, "GBps": gb / ms * 1e3, "frac_of_8TBps": gb / ms * 1e3 / 8000,
                              "Gexp_per_s": 2 * dim * N * L / ms / 1e6}
print(json.dumps(out))
