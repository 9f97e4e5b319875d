#!/usr/bin/env python3
"""Can a whole BASELINE config-3 training step (forward, L1 + 0.1 FFT-L1, backward, AdamW) be captured into a HIP graph, and what does
the replay cost against the eager step?  python tools/try_graph_train.py [steps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wave_mamba_amd as wm
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1234)
lq, gt = torch.rand(8, 3, 512, 512, generator=g).to(dev), torch.rand(8, 3, 512, 512, generator=g).to(dev)


def fresh():
    torch.manual_seed(0)
    net = wm.WaveMamba(**bench.SHIPPED).train().to(dev)
    params = [p for p in net.parameters() if p.requires_grad]
    return net, torch.optim.AdamW(params, lr=5e-4, weight_decay=1e-3, betas=(0.9, 0.99), fused=True, capturable=True)


def step(net, opt):
    opt.zero_grad(set_to_none=True)
    out = net(lq)
    l_pix, l_freq = wm.trainer.losses(out, gt)
    (l_pix + l_freq).mean().backward()
    opt.step()
    return l_pix.detach(), l_freq.detach()


# eager reference: 3 + steps steps
net, opt = fresh()
for _ in range(3):
    step(net, opt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    le = step(net, opt)
torch.cuda.synchronize()
t_eager = (time.perf_counter() - t0) / steps
le = [float(v) for v in le]
del net, opt

net, opt = fresh()
side = torch.cuda.Stream(dev)
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        step(net, opt)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
try:
    with torch.cuda.graph(graph):
        out = net(lq)
        l_pix, l_freq = wm.trainer.losses(out, gt)
        (l_pix + l_freq).mean().backward()
        opt.step()
except Exception as e:
    print("capture FAILED:", type(e).__name__, str(e)[:400])
    sys.exit(1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    graph.replay()
torch.cuda.synchronize()
t_graph = (time.perf_counter() - t0) / steps
lg = [float(l_pix), float(l_freq)]
print(f"eager {t_eager * 1e3:.2f} ms per step, graph replay {t_graph * 1e3:.2f} ms per step")
print(f"losses after 3 + {steps} steps: eager {le}, graphed {lg}, rel diff {max(abs(a - b) / abs(a) for a, b in zip(le, lg)):.2e}")
