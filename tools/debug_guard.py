#!/usr/bin/env python3
"""Uninitialised reads and out-of-bounds writes of the inference forward, at the size of record (UHD, fp32 and bf16 planes).

Every `torch.empty` / `torch.empty_like` issued by wave_mamba_amd.ops is replaced by an allocation with a 4-KB guard zone on
both sides, the WHOLE buffer (guards + payload) filled with 0xFF bytes (fp32 / bf16: NaN, int32: -1) on the allocating stream
before it is handed out, and kept alive until the check:
  * a kernel that consumes memory it has not written puts NaN into its output - every op's returned tensors are checked, the
    first op (name, shapes) is reported;
  * a kernel that writes outside its output leaves a non-0xFF byte in a guard zone - reported with the allocating call.
MODE (env): single | multi ; PLANES: f32 | bf16 ; HW: 2176x3840
"""
import os, sys, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wave_mamba_amd as wm
import bench

GUARD = 4096


class GuardTorch:
    def __init__(self, real):
        self._real = real
        self.live = []
        self.on = True

    def __getattr__(self, n):
        return getattr(self._real, n)

    def empty(self, *size, dtype=None, device=None, **kw):
        real = self._real
        if len(size) == 1 and isinstance(size[0], (tuple, list, real.Size)):
            size = tuple(size[0])
        if not self.on or device is None or real.device(device).type != "cuda":
            return real.empty(size, dtype=dtype, device=device, **kw)
        dtype = dtype or real.float32
        n = 1
        for s in size:
            n *= int(s)
        es = real.empty(0, dtype=dtype).element_size()
        g = GUARD // es
        buf = real.empty(n + 2 * g, dtype=dtype, device=device)
        buf.view(real.uint8).fill_(0xFF)
        where = " <- ".join(f"{f.name}:{f.lineno}" for f in reversed(traceback.extract_stack(limit=5)[:-1]))
        self.live.append((buf, g, n, where, tuple(size), dtype))
        return buf[g:g + n].view(size)

    def empty_like(self, x, dtype=None, device=None, **kw):
        return self.empty(tuple(x.shape), dtype=dtype or x.dtype, device=device or x.device)

    def check(self):
        self._real.cuda.synchronize()
        bad = 0
        for buf, g, n, where, size, dtype in self.live:
            u8 = buf.view(self._real.uint8)
            es = buf.element_size()
            lo, hi = u8[:g * es], u8[(g + n) * es:]
            nlo, nhi = int((lo != 0xFF).sum()), int((hi != 0xFF).sum())
            if nlo or nhi:
                bad += 1
                print(f"  GUARD HIT: {size} {dtype} allocated at {where}: {nlo} bytes below, {nhi} bytes above the payload", flush=True)
                if nhi:
                    idx = (hi != 0xFF).nonzero().flatten()
                    print(f"     above: first offset {int(idx[0])}, last {int(idx[-1])}")
                if nlo:
                    idx = (lo != 0xFF).nonzero().flatten()
                    print(f"     below: first offset {int(idx[0])}, last {int(idx[-1])} of {g * es}")
        print(f"  guard zones checked: {len(self.live)} allocations, {bad} with hits", flush=True)
        self.live.clear()
        return bad


def tensors_of(o):
    if isinstance(o, torch.Tensor):
        yield o
    elif isinstance(o, (tuple, list)):
        for v in o:
            yield from tensors_of(v)


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    net = wm.WaveMamba(**bench.SHIPPED).eval().to(dev)
    unet = net.restoration_network
    g = torch.Generator().manual_seed(1234)
    H, W = (int(v) for v in os.environ.get("HW", "2176x3840").split("x"))
    x = torch.rand(1, 3, H, W, generator=g).to(dev)
    planes = {"f32": torch.float32, "bf16": torch.bfloat16}[os.environ.get("PLANES", "bf16")]
    multi = os.environ.get("MODE", "single") == "multi"
    check_ops = os.environ.get("CHECK_OPS", "1") == "1"
    deferred = os.environ.get("CHECK_OPS", "1") == "2"          # non-finite counts queued on the op's own stream, read after the forward
    real_get = wm.ops.get_plane_dtype
    with torch.no_grad():
        wm.ops.set_plane_dtype(planes)
        unet.two_streams = False
        base = unet(x); torch.cuda.synchronize()
        print(f"{H}x{W} planes {planes} multi {multi}: clean single-stream forward finite: {bool(torch.isfinite(base).all())}", flush=True)
        unet.two_streams = multi
        if multi:
            wm.ops.get_plane_dtype = lambda: torch.float32      # only UNet.forward's stream-order switch reads it
        gt = GuardTorch(torch)
        wm.ops.torch = gt
        first = []
        pending = []
        if check_ops or deferred:
            names = [n for n in dir(wm.ops) if not n.startswith("_") and callable(getattr(wm.ops, n))
                     and getattr(getattr(wm.ops, n), "__module__", "") == wm.ops.__name__ and not isinstance(getattr(wm.ops, n), type)]
            skip = {"set_plane_dtype", "get_plane_dtype", "prof_enable", "prof_collect", "conv2d_cache_clear", "conv2d_select"}
            orig = {n: getattr(wm.ops, n) for n in names if n not in skip and not n.endswith("_supported")}

            def wrap(n, f):
                def w(*a, **k):
                    o = f(*a, **k)
                    if deferred:
                        for i, t in enumerate(tensors_of(o)):
                            if t.is_floating_point():
                                pending.append((n, i, tuple(t.shape), torch.cuda.current_stream().cuda_stream,
                                                (~torch.isfinite(t)).sum()))
                        return o
                    if not first:
                        for i, t in enumerate(tensors_of(o)):
                            if t.is_floating_point():
                                torch.cuda.synchronize()
                                if not bool(torch.isfinite(t.float()).all()):
                                    nbad = int((~torch.isfinite(t.float())).sum())
                                    first.append(n)
                                    ins = [tuple(v.shape) for v in tensors_of(a)]
                                    print(f"  FIRST NON-FINITE OUTPUT: op {n}, output {i} {tuple(t.shape)} {t.dtype}: {nbad} elements; "
                                          f"inputs {ins[:4]}", flush=True)
                                    bi = (~torch.isfinite(t.float())).nonzero()
                                    print(f"     first bad index {bi[0].tolist()}, last {bi[-1].tolist()}", flush=True)
                                    break
                    return o
                return w
            for n, f in orig.items():
                setattr(wm.ops, n, wrap(n, f))
        for rep in range(2):
            out = unet(x); torch.cuda.synchronize()
            fin = bool(torch.isfinite(out).all())
            print(f"  poisoned-allocation forward {rep}: finite {fin}, max |diff| to the clean forward "
                  f"{float((out - base).abs().nan_to_num(1e9).max()):.3e}", flush=True)
            gt.check()
            if deferred:
                seen = set()
                for k, (n, i, shape, st, cnt) in enumerate(pending):
                    c = int(cnt)
                    if c and (st, n) not in seen:
                        seen.add((st, n))
                        print(f"    op #{k} {n} output {i} {shape} on stream {st:#x}: {c} non-finite", flush=True)
                pending.clear()
        wm.ops.torch = torch
        wm.ops.get_plane_dtype = real_get
        wm.ops.set_plane_dtype(torch.float32)


if __name__ == "__main__":
    main()
