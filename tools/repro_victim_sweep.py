#!/usr/bin/env python3
"""Which kernels of the inference forward give different results while a 3x3 matrix-core convolution runs on another stream?
(tools/repro_pk_lanes.py found dwconv3x3<bf16>.)  Every victim is one operator group of the shipped network on fixed inputs,
launched N times on the main stream, each launch compared bit for bit with the launch that ran alone; the aggressor is the
library's 3x3 convolution (64 -> 64) one level larger, launched once per victim launch on a second stream.  Nothing is shared
between the streams.
env: N=200  LEVELS=3,2  AGGRESSOR=conv3x3|firstgen|none
"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wave_mamba_amd as wm
import bench

dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = wm.WaveMamba(**bench.SHIPPED).eval().to(dev)
unet = net.restoration_network
g = torch.Generator().manual_seed(11)
N = int(os.environ.get("N", "200"))
LEVELS = [int(v) for v in os.environ.get("LEVELS", "3,2").split(",")]
DIMS = {1: (1088, 1920), 2: (544, 960), 3: (272, 480)}
side = torch.cuda.Stream(device=dev)


def rnd(*shape):
    return torch.randn(*shape, generator=g).to(dev)


def bits(t):
    return t.view(torch.int16) if t.dtype == torch.bfloat16 else t


def flat(o):
    return [o] if isinstance(o, torch.Tensor) else [t for v in o for t in flat(v)]


def victims(level):
    H, W = DIMS[level]
    dg = getattr(unet, f"down_group{level}")
    ug = getattr(unet, f"up_group{level}")
    blk = dg.l_blk[0]
    x32 = rnd(1, 32, H, W)
    x64 = rnd(1, 64, H, W)
    low = rnd(1, 32, H, W)
    hl, lh, hh = rnd(1, 32, H, W), rnd(1, 32, H, W), rnd(1, 32, H, W)
    x96 = rnd(1, 96, H, W)
    full = rnd(1, 32, 2 * H, 2 * W)
    ss = blk.self_attention

    def lfss(dt):
        def f():
            prev = wm.ops.set_plane_dtype(dt)
            try:
                return wm.ops.lfss_block_forward(x32, (H, W), blk, tok_nchw=True, out_nchw=True)
            finally:
                wm.ops.set_plane_dtype(prev)
        return f
    xb = x64.bfloat16()
    v = {
        "lfss_block fp32 planes": lfss(torch.float32),
        "lfss_block bf16 planes": lfss(torch.bfloat16),
        "dwconv3x3+silu fp32": lambda: wm.ops.dwconv3x3(x64, ss.conv2d.weight, ss.conv2d.bias, "silu"),
        "dwconv3x3+silu bf16": lambda: wm.ops.dwconv3x3(xb, ss.conv2d.weight, ss.conv2d.bias, "silu"),
        "dwconv3x3 96ch fp32 (qkv)": lambda: wm.ops.dwconv3x3(x96, dg.h_blk[0].attn.qkv_dwconv.weight, dg.h_blk[0].attn.qkv_dwconv.bias, "none"),
        "dwconv3x3+gelu fp32": lambda: wm.ops.dwconv3x3(x32, dg.h_blk[0].ffn.project_out[0].weight, dg.h_blk[0].ffn.project_out[0].bias, "gelu"),
        "ss2d_core fp32": lambda: wm.ops.ss2d_core(x64, ss.x_proj_weight, ss.dt_projs_weight, ss.dt_projs_bias, ss.A_logs, ss.Ds),
        "ss2d_core bf16": lambda: wm.ops.ss2d_core(xb, ss.x_proj_weight, ss.dt_projs_weight, ss.dt_projs_bias, ss.A_logs, ss.Ds),
        "HFEBlock": lambda: dg.h_blk[0](x32, low),
        "SKFF": lambda: dg.h_fusion([hl, lh, hh]),
        "dwt": lambda: wm.ops.dwt_init(full),
        "iwt pair": lambda: wm.ops.iwt_init_pair(x32, x96),
        "conv3x3 64->32 (l_conv, cat)": lambda: wm.ops.conv2d(x32, dg.l_conv.weight, dg.l_conv.bias, low),
        "conv3x3 32->96 (h_out_conv)": lambda: wm.ops.conv2d(x32, ug.h_out_conv.weight, ug.h_out_conv.bias),
        "conv1x1+LN 32->96 (qkv)": lambda: wm.ops.conv2d_ln(x32, dg.h_blk[0].norm1.weight, dg.h_blk[0].norm1.bias, dg.h_blk[0].norm1.eps,
                                                            dg.h_blk[0].attn.qkv.weight, dg.h_blk[0].attn.qkv.bias),
        "layernorm2d": lambda: wm.ops.layernorm2d(x32, dg.h_blk[0].LayerNorm.weight, dg.h_blk[0].LayerNorm.bias, dg.h_blk[0].LayerNorm.eps),
        "gram": lambda: wm.ops.gram(x32.flatten(2), low.flatten(2)),
    }
    return v


def aggressor(level, kind):
    if kind == "none":
        return None
    H, W = DIMS[max(1, level - 1)]
    xa = rnd(1, 64, H, W)
    w3 = rnd(64, 64, 3, 3) / 24

    def run():
        if kind == "firstgen":
            wm.ops.conv2d_select(wm.ops.CONV3X3_FIRST_GEN)
        try:
            return wm.ops.conv2d(xa, w3)
        finally:
            if kind == "firstgen":
                wm.ops.conv2d_select(wm.ops.CONV3X3_AUTO)
    return run


with torch.no_grad():
    print(f"library build {wm._lib.load().wm_build_id().decode()}", flush=True)
    for level in LEVELS:
        for kind in os.environ.get("AGGRESSOR", "conv3x3,firstgen").split(","):
            a = aggressor(level, kind)
            for name, v in victims(level).items():
                ref = [t.clone() for t in flat(v())]; torch.cuda.synchronize()
                again = flat(v()); torch.cuda.synchronize()
                alone_ok = all(torch.equal(bits(p), bits(q)) for p, q in zip(ref, again))
                cnts, keep = [], []
                for i in range(N):
                    if a is not None:
                        with torch.cuda.stream(side):
                            keep.append(a())
                            if len(keep) > 6:
                                keep.pop(0)
                    o = flat(v())
                    cnts.append(sum((bits(p) != bits(q)).sum() for p, q in zip(o, ref)))
                torch.cuda.synchronize()
                bad = [int(c) for c in cnts]
                nbad = sum(1 for c in bad if c)
                print(f"level {level} aggressor {kind:9s} victim {name:32s}: {nbad:4d} of {N} launches differ"
                      + ("" if alone_ok else "   [NOT reproducible alone]")
                      + (f"   elements {sorted(set(c for c in bad if c))[:6]}" if nbad else ""), flush=True)
