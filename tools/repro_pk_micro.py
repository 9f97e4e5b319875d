#!/usr/bin/env python3
"""Driver of tools/ubench_pk_coexec.hip: dependent packed-fp32 chains (victim, main stream) against aggressor kernels on a
second stream.  Prints mismatch counts per (victim variant, aggressor) and the lanes / halves they fall on."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wave_mamba_amd as wm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "build", "ubench_pk_coexec.so"))
P, I = ctypes.c_void_p, ctypes.c_int
lib.pk_victim_launch.argtypes = [I, I, P, I, I, I, P, P]
lib.aggr_launch.argtypes = [I, P, I, P, I, I, P]
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(3)
REPS = int(os.environ.get("REPS", "20"))
BLOCKS, ITERS = int(os.environ.get("BLOCKS", "2048")), int(os.environ.get("ITERS", "1000"))

counts = torch.zeros(128, dtype=torch.int32, device=dev)
sink = torch.zeros(1024, device=dev)
src = torch.randint(0, 2 ** 31 - 1, (16 * 1024 * 1024,), dtype=torch.int32, device=dev)      # 64 MB
xa = torch.randn(1, 64, 544, 960, generator=g).to(dev)
w3 = (torch.randn(64, 64, 3, 3, generator=g) / 24).to(dev)
w1 = (torch.randn(64, 64, 1, 1, generator=g) / 8).to(dev)
side = torch.cuda.Stream(device=dev)


def st():
    return torch.cuda.current_stream().cuda_stream


keep = []
AGGR = {
    "none": None,
    "mfma loop": lambda: lib.aggr_launch(0, None, 0, sink.data_ptr(), 256, 3000, st()),
    "lds-dma loop": lambda: lib.aggr_launch(1, src.data_ptr(), src.numel() // 4, sink.data_ptr(), 256, 3000, st()),
    "lds-read + mfma loop (random data)": lambda: lib.aggr_launch(3, src.data_ptr(), src.numel() // 4, sink.data_ptr(), 256, 3000, st()),
    "valu loop": lambda: lib.aggr_launch(2, None, 0, sink.data_ptr(), 256, 300, st()),
    "lib conv3x3 (wave-specialised)": lambda: keep.append(wm.ops.conv2d(xa, w3)),
    "lib conv1x1": lambda: keep.append(wm.ops.conv2d(xa, w1)),
}
MODES = {0: "pk, SGPR weights", 1: "pk, VGPR weights", 3: "pk, v_mov_b64 start", 2: "plain v_fma_f32",
         4: "pk + dwordx2 load in flight", 5: "pk + ushort load in flight", 6: "pk + ds_bpermute in flight", 8: "pk + dword load in flight", 9: "pk VGPR wts + ushort load",
         10: "pk SGPR wts + op_sel swap", 11: "pk VGPR wts + op_sel swap",
         12: "pk SGPR wts + src1 (lo,lo)", 13: "pk const 2.0 + op_sel swap",
         14: "pk_add VGPR only, src1 swapped", 15: "pk_mul VGPR only, op_sel:[1,0]"}
if os.environ.get("MODES"):
    MODES = {int(m): MODES[int(m)] for m in os.environ["MODES"].split(",")}
GAPS = {0: "gap 0", 1: "gap 1 VALU", 2: "gap 2 VALU", 3: "gap 3 VALU", 4: "gap s_nop 0"}
sel_aggr = os.environ.get("AGGR")
with torch.no_grad():
    for an, a in AGGR.items():
        if sel_aggr and not any(t in an for t in sel_aggr.split(",")):
            continue
        for mode, mn in MODES.items():
            for gap, gn in GAPS.items():
                if mode == 2 and gap == 4:
                    continue
                counts.zero_(); torch.cuda.synchronize()
                for r in range(REPS):
                    if a is not None:
                        with torch.cuda.stream(side):
                            a()
                            if len(keep) > 8:
                                keep.pop(0)
                    rc = lib.pk_victim_launch(mode, gap, counts.data_ptr(), BLOCKS, ITERS, r, st(), src.data_ptr())
                    assert rc == 0, rc
                torch.cuda.synchronize()
                c = counts.cpu().tolist()
                lo, hi = c[:64], c[64:]
                tot = sum(c)
                msg = ""
                if tot:
                    msg = (f"  low-half lanes {[i for i in range(64) if lo[i]][:20]} ({sum(lo)}), "
                           f"high-half lanes {[i for i in range(64) if hi[i]][:20]} ({sum(hi)})")
                print(f"aggressor {an:32s} victim {mn:28s} {gn:12s}: {tot:8d} wrong of {REPS * BLOCKS * 256 * ITERS:.2e} chains{msg}",
                      flush=True)
