#!/usr/bin/env python3
"""Which kernel's output differs first when the UHD forward runs in the multi-stream order?

Pass 1 (reference): the multi-stream code path with every fork / join replaced by a device synchronisation (bit-equal to the
single-stream order, tools/debug_race_forks.py variant B) - every op's outputs are kept, in call order.
Pass 2..: the same code path as shipped (events, overlap): every op's outputs are compared with pass 1's on the op's own
stream (a count of differing elements + the largest difference, queued behind the op, read after the forward).
Prints, per stream, the first ops whose outputs differ.
env: PLANES=bf16|f32, HW=2176x3840, REPS=3
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
g = torch.Generator().manual_seed(1234)
H, W = (int(v) for v in os.environ.get("HW", "2176x3840").split("x"))
x = torch.rand(1, 3, H, W, generator=g).to(dev)
planes = {"f32": torch.float32, "bf16": torch.bfloat16}[os.environ.get("PLANES", "bf16")]
REPS = int(os.environ.get("REPS", "3"))


def tensors_of(o):
    if isinstance(o, torch.Tensor):
        yield o
    elif isinstance(o, (tuple, list)):
        for v in o:
            yield from tensors_of(v)


names = [n for n in dir(wm.ops) if not n.startswith("_") and callable(getattr(wm.ops, n))
         and getattr(getattr(wm.ops, n), "__module__", "") == wm.ops.__name__ and not isinstance(getattr(wm.ops, n), type)]
skip = {"set_plane_dtype", "get_plane_dtype", "prof_enable", "prof_collect", "conv2d_cache_clear", "conv2d_select"}
orig = {n: getattr(wm.ops, n) for n in names if n not in skip and not n.endswith("_supported")}
state = {"mode": None, "k": 0, "depth": 0}
ref, pending = [], []


def wrap(n, f):
    def w(*a, **k):
        state["depth"] += 1
        try:
            o = f(*a, **k)
        finally:
            state["depth"] -= 1
        outs = [t for t in tensors_of(o)]
        kk = state["k"]; state["k"] += 1
        if state["mode"] == "record":
            ref.append((n, state["depth"], outs))
        elif state["mode"] == "compare":
            rn, rd, routs = ref[kk]
            assert rn == n and len(routs) == len(outs), (kk, rn, n)
            st = torch.cuda.current_stream().cuda_stream
            for i, (t, r) in enumerate(zip(outs, routs)):
                tf, rf = (t.float(), r.float()) if t.is_floating_point() else (t, r)
                pending.append((kk, n, state["depth"], i, tuple(t.shape), str(t.dtype), st, (tf != rf).sum(),
                                (tf - rf).abs().max() if t.is_floating_point() else (tf != rf).sum()))
        return o
    return w


for n, f in orig.items():
    setattr(wm.ops, n, wrap(n, f))


class CaptureTorch:
    """every tensor ops.py allocates, in allocation order, with the allocating call sites"""
    def __init__(self, real):
        self._real, self.live = real, []

    def __getattr__(self, n):
        return getattr(self._real, n)

    def _note(self, t):
        if t.is_cuda:
            import traceback
            where = " <- ".join(f"{f.name}:{f.lineno}" for f in reversed(traceback.extract_stack(limit=5)[:-2]))
            self.live.append((where, t))
        return t

    def empty(self, *a, **k):
        return self._note(self._real.empty(*a, **k))

    def empty_like(self, *a, **k):
        return self._note(self._real.empty_like(*a, **k))


CAPTURE = os.environ.get("CAPTURE", "0") == "1"
cap = CaptureTorch(torch)
if CAPTURE:
    wm.ops.torch = cap
ref_allocs = []

with torch.no_grad():
    wm.ops.set_plane_dtype(planes)
    unet.two_streams = True
    wm.ops.get_plane_dtype = lambda: torch.float32          # (only UNet.forward's stream-order switch reads it)
    ws, we = torch.cuda.Stream.wait_stream, torch.cuda.Stream.wait_event
    torch.cuda.Stream.wait_stream = lambda self, other: torch.cuda.synchronize()
    torch.cuda.Stream.wait_event = lambda self, ev: torch.cuda.synchronize()
    state["mode"] = None
    unet(x); torch.cuda.synchronize()                       # warm-up (caches, side streams)
    state.update(mode="record", k=0)
    cap.live.clear()
    base = unet(x); torch.cuda.synchronize()
    ref_allocs = list(cap.live); cap.live.clear()
    torch.cuda.Stream.wait_stream, torch.cuda.Stream.wait_event = ws, we
    main = torch.cuda.current_stream().cuda_stream
    print(f"{H}x{W} {planes}: {len(ref)} op calls recorded; main stream {main:#x}", flush=True)
    for rep in range(REPS):
        state.update(mode="compare", k=0)
        cap.live.clear()
        out = unet(x); torch.cuda.synchronize()
        print(f"pass {rep}: output max |diff| {float((out - base).abs().max()):.3e}", flush=True)
        shown = {}
        for kk, n, depth, i, shape, dt, st, cnt, mx in pending:
            c = int(cnt)
            if c:
                s = shown.setdefault(st, 0)
                if s < 6:
                    shown[st] = s + 1
                    print(f"   call #{kk:3d} {'  ' * depth}{n} out {i} {shape} {dt} on {'main' if st == main else hex(st)}: "
                          f"{c} elements differ, max |diff| {float(mx):.3e}", flush=True)
        pending.clear()
        if CAPTURE:
            assert len(cap.live) == len(ref_allocs), (len(cap.live), len(ref_allocs))
            nshown = 0
            for j, ((where, t), (rwhere, r)) in enumerate(zip(cap.live, ref_allocs)):
                if t.dtype == torch.uint8 or t.shape != r.shape:
                    continue                                    # workspaces: unwritten parts are whatever the block held
                ne = (t != r)
                c = int(ne.sum())
                if c and nshown < 12:
                    nshown += 1
                    idx = ne.nonzero()
                    print(f"   alloc #{j} {tuple(t.shape)} {t.dtype} [{where}]: {c} elements differ, max |diff| "
                          f"{float((t.float() - r.float()).abs().max()):.3e}", flush=True)
                    lo, hi = idx.min(0).values.tolist(), idx.max(0).values.tolist()
                    print(f"        index range {lo} .. {hi}; first {idx[:6].tolist()}", flush=True)
                    if idx.shape[1] == 4:
                        rows = torch.unique(idx[:, 2]).tolist(); cols = torch.unique(idx[:, 3]).tolist(); chs = torch.unique(idx[:, 1]).tolist()
                        print(f"        channels {chs[:40]}{' ...' if len(chs) > 40 else ''}\n        rows {rows[:40]}{' ...' if len(rows) > 40 else ''}\n"
                              f"        cols {cols[:70]}{' ...' if len(cols) > 70 else ''}", flush=True)
    wm.ops.set_plane_dtype(torch.float32)
