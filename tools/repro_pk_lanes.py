#!/usr/bin/env python3
"""Minimal reproducer of the multi-stream mismatch (round 5): ONE kernel with compiler-generated packed-fp32 arithmetic
(dwconv3x3 + SiLU on bf16 planes: `v_pk_fma_f32 ... op_sel`, `v_pk_add_f32 ... op_sel_hi`) run repeatedly on fixed inputs on
one stream while another stream keeps the chip busy with a second kernel.  No data is shared between the streams, every
buffer is distinct and alive.  Counts the victim launches whose output differs from the launch that ran alone.

env: VICTIM=dwconv_bf16|dwconv_f32|lfss_bf16 ...   AGGRESSOR=conv3x3|conv1x1|dwconv_f32|haar|none   N=300
"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wave_mamba_amd as wm

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(7)
N = int(os.environ.get("N", "300"))
H, W = (int(v) for v in os.environ.get("HW", "272x480").split("x"))


def victims():
    xb = torch.randn(1, 64, H, W, generator=g).to(dev)
    wgt = (torch.randn(64, 1, 3, 3, generator=g) / 3).to(dev)
    b = torch.randn(64, generator=g).to(dev)
    xbf = xb.bfloat16()
    return {
        "dwconv_bf16_silu": lambda: wm.ops.dwconv3x3(xbf, wgt, b, "silu"),
        "dwconv_bf16_none": lambda: wm.ops.dwconv3x3(xbf, wgt, b, "none"),
        "dwconv_f32_silu": lambda: wm.ops.dwconv3x3(xb, wgt, b, "silu"),
    }


def aggressors():
    xa = torch.randn(1, 64, 544, 960, generator=g).to(dev)
    w3 = (torch.randn(64, 64, 3, 3, generator=g) / 24).to(dev)
    w1 = (torch.randn(64, 64, 1, 1, generator=g) / 8).to(dev)
    wd = (torch.randn(64, 1, 3, 3, generator=g) / 3).to(dev)
    xs = torch.randn(1, 32, 544, 960, generator=g).to(dev)
    import ctypes
    ub = None
    ubp = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "ubench_pk_coexec.so")
    if os.path.exists(ubp):
        ub = ctypes.CDLL(ubp)
        ub.aggr_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    sink = torch.zeros(1024, device=dev)
    src = torch.randint(0, 2 ** 31 - 1, (16 * 1024 * 1024,), dtype=torch.int32, device=dev)
    st = lambda: torch.cuda.current_stream().cuda_stream
    syn = {} if ub is None else {
        "syn_mfma": lambda: ub.aggr_launch(0, None, 0, sink.data_ptr(), 256, 400, st()),
        "syn_ldsdma": lambda: ub.aggr_launch(1, src.data_ptr(), src.numel() // 4, sink.data_ptr(), 256, 400, st()),
        "syn_valu": lambda: ub.aggr_launch(2, None, 0, sink.data_ptr(), 256, 40, st()),
    }
    xg = torch.randn(1, 64, 544, 960, generator=g).to(dev)
    w3g = (torch.randn(64, 64, 3, 3, generator=g) / 24).to(dev)
    w1g = (torch.randn(64, 64, 1, 1, generator=g) / 8).to(dev)
    b1g = torch.randn(64, generator=g).to(dev)
    w32 = (torch.randn(32, 64, 3, 3, generator=g) / 24).to(dev)

    def conv_first_gen():
        wm.ops.conv2d_select(wm.ops.CONV3X3_FIRST_GEN)
        try:
            return wm.ops.conv2d(xa, w3)
        finally:
            wm.ops.conv2d_select(wm.ops.CONV3X3_AUTO)
    return {
        **syn,
        "conv3x3_firstgen": conv_first_gen,
        "conv3x3_64to32": lambda: wm.ops.conv2d(xa, w32),
        "conv_gated": lambda: wm.ops.conv2d_gated(xg, w3g, w1g, b1g),
        "conv3x3": lambda: wm.ops.conv2d(xa, w3),
        "conv1x1": lambda: wm.ops.conv2d(xa, w1),
        "dwconv_f32": lambda: wm.ops.dwconv3x3(xa, wd, None, "none"),
        "haar": lambda: wm.ops.dwt_init(xs),
        "none": None,
    }


def main():
    vs, ags = victims(), aggressors()
    vsel = os.environ.get("VICTIM", ",".join(vs)).split(",")
    asel = os.environ.get("AGGRESSOR", ",".join(ags)).split(",")
    side = torch.cuda.Stream(device=dev)
    main_st = torch.cuda.current_stream(dev)
    with torch.no_grad():
        for vn in vsel:
            v = vs[vn]
            ref = v(); torch.cuda.synchronize()
            again = v(); torch.cuda.synchronize()
            assert torch.equal(ref.view(torch.int16) if ref.dtype == torch.bfloat16 else ref,
                               again.view(torch.int16) if again.dtype == torch.bfloat16 else again), "not reproducible ALONE"
            for an in asel:
                a = ags[an]
                if a is not None:
                    for _ in range(3):
                        a()
                    torch.cuda.synchronize()
                cnts, outs = [], []
                for i in range(N):
                    if a is not None:
                        with torch.cuda.stream(side):
                            keep = a()                      # (kept alive below: no allocator reuse across streams)
                            outs.append(keep)
                    o = v()
                    cnts.append((o.float() != ref.float()).sum())
                    if len(outs) > 8:
                        outs.pop(0)
                torch.cuda.synchronize()
                bad = [int(c) for c in cnts]
                nbad = sum(1 for c in bad if c)
                print(f"victim {vn:18s} aggressor {an:10s}: {nbad} of {N} launches differ"
                      + (f" (elements per bad launch: {sorted(set(c for c in bad if c))[:8]})" if nbad else ""), flush=True)
                if nbad and os.environ.get("DETAIL", "1") == "1":
                    # one more differing launch, looked at closely
                    for _ in range(200):
                        if a is not None:
                            with torch.cuda.stream(side):
                                keep = a()
                        o = v(); torch.cuda.synchronize()
                        ne = (o.float() != ref.float())
                        if bool(ne.any()):
                            idx = ne.nonzero()
                            cols = torch.unique(idx[:, 3]).tolist()
                            print(f"    differing: channels {torch.unique(idx[:, 1]).tolist()[:10]} rows {torch.unique(idx[:, 2]).tolist()[:10]} "
                                  f"cols {cols[:40]}", flush=True)
                            lanes = sorted(set((c // 4) % 64 for c in cols)); elems = sorted(set(c % 4 for c in cols))
                            print(f"    lanes of the wave {lanes}; elements of the quad {elems}", flush=True)
                            for j in idx[:6].tolist():
                                print(f"      [{j[1]},{j[2]},{j[3]}] got {float(o[tuple(j)]):+.6f} expected {float(ref[tuple(j)]):+.6f}", flush=True)
                            break


if __name__ == "__main__":
    main()
