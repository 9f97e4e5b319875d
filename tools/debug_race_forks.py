#!/usr/bin/env python3
"""bf16 planes + multi-stream order at UHD: what kind of hazard is it?
  A  as is
  B  every Stream.wait_stream / wait_event replaced by a DEVICE synchronisation (forks and joins become exact; the overlap of a
     level's side-stream branch with the main stream's next levels stays) - still differing => interference between kernels that
     run concurrently (stray writes, uninitialised reads), gone => a missing / broken dependency
  E  every tensor ops.py allocates kept alive until the end of the forward (no allocator reuse inside one forward)
  F  E + A's stream order but the allocations zero-filled
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
REPS = int(os.environ.get("REPS", "4"))


class KeepTorch:
    def __init__(self, real, zero=False):
        self._real, self.live, self.zero = real, [], zero

    def __getattr__(self, n):
        return getattr(self._real, n)

    def empty(self, *a, **k):
        t = self._real.empty(*a, **k)
        if t.is_cuda:
            self.live.append(t)
            if self.zero:
                t.view(self._real.uint8).zero_() if t.is_contiguous() else t.zero_()
        return t

    def empty_like(self, *a, **k):
        t = self._real.empty_like(*a, **k)
        if t.is_cuda:
            self.live.append(t)
            if self.zero:
                t.zero_()
        return t


def run(label):
    d = []
    for _ in range(REPS):
        o = unet(x); torch.cuda.synchronize(); d.append(float((o - base).abs().max()))
        if isinstance(wm.ops.torch, KeepTorch):
            wm.ops.torch.live.clear()
    print(f"{H}x{W} {planes} {label}: max |diff| vs single-stream {['%.2e' % v for v in d]}", flush=True)


with torch.no_grad():
    wm.ops.set_plane_dtype(planes)
    unet.two_streams = False
    base = unet(x); torch.cuda.synchronize()
    b2 = unet(x); torch.cuda.synchronize()
    print("single-stream reproducible:", bool(torch.equal(base, b2)), flush=True)
    unet.two_streams = True
    real_get = wm.ops.get_plane_dtype
    wm.ops.get_plane_dtype = lambda: torch.float32          # (only UNet.forward's stream-order switch reads it)
    run("A as is")
    ws, we = torch.cuda.Stream.wait_stream, torch.cuda.Stream.wait_event
    torch.cuda.Stream.wait_stream = lambda self, other: torch.cuda.synchronize()
    torch.cuda.Stream.wait_event = lambda self, ev: torch.cuda.synchronize()
    run("B device sync at every fork / join")
    def host_wait_stream(self, other):
        other.synchronize()                      # the host waits for `other` alone; streams not involved keep running
    def host_wait_event(self, ev):
        ev.synchronize()
    torch.cuda.Stream.wait_stream, torch.cuda.Stream.wait_event = host_wait_stream, host_wait_event
    run("B' host waits for the awaited stream / event only (overlap with the other streams kept)")
    torch.cuda.Stream.wait_stream, torch.cuda.Stream.wait_event = ws, we
    wm.ops.torch = KeepTorch(torch)
    run("E allocations kept alive")
    wm.ops.torch = KeepTorch(torch, zero=True)
    run("F allocations kept alive + zero-filled")
    wm.ops.torch = torch
    run("A again")
    wm.ops.get_plane_dtype = real_get
    wm.ops.set_plane_dtype(torch.float32)
