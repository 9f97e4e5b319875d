"""Reproducer (round 6): an accumulate-into operator of the library captured into a HIP graph, replayed with eager GPU work between
the replays.  With the library zeroing through hipMemsetAsync (builds before 540a1c8ca56bafa3) the second replay returns a buffer in
which every fourth float is the low half of the address of one of the eager temporaries: the memset NODE fills with a 16-byte pattern
it re-reads from recycled memory.  The library zeroes with a kernel now and this prints `0 bad` for every replay.
    python tools/repro_graph_memset_node.py        (GPU box)"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wave_mamba_amd as wm
from wave_mamba_amd import ops, _lib
dev = torch.device("cuda", 0)
lib = _lib.load()
B, C, H, W = 2, 32, 32, 32
x = torch.randn(B, C, H, W, device=dev); gy = torch.randn(B, C, H, W, device=dev)
def wgrad(buf):
    dW, db = buf[:9 * C], buf[9 * C:]
    ops.check(lib.wm_dwconv3x3_wgrad(x.data_ptr(), gy.data_ptr(), dW.data_ptr(), db.data_ptr(), B, C, H, W, torch.cuda.current_stream().cuda_stream), "wgrad")
ref = torch.empty(10 * C, device=dev); wgrad(ref); torch.cuda.synchronize()
ref_cpu = ref.cpu().clone()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    buf = torch.empty(10 * C, device=dev)
    wgrad(buf)
print("buf %x ref %x x %x gy %x" % (buf.data_ptr(), ref.data_ptr(), x.data_ptr(), gy.data_ptr()))
for r in range(3):
    gr.replay(); torch.cuda.synchronize()
    bc = buf.cpu()
    idx = ((bc - ref_cpu).abs() > 1e-3).nonzero().flatten().tolist()
    print(f"replay {r}: {len(idx)} bad; idx[:40] {idx[:40]}")
    if idx:
        print("    diffs", [(i, float(bc[i] - ref_cpu[i])) for i in idx[:16]])
    # the GPU-side check of the failing scripts (temporaries from the regular pool)
    t1 = buf - ref; t2 = t1.abs(); t3 = t2 > 1e-3; t4 = t3.sum(); t5 = t2.argmax()
    torch.cuda.synchronize()
    print("    temporaries at", [hex(t.data_ptr()) for t in (t1, t2, t3, t4, t5)], "gpu check says", int(t4))
    pre = buf.cpu()
    print("    buf unchanged by the check:", bool(torch.equal(pre, bc)))
    del t1, t2, t3, t4, t5
