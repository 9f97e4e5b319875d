#!/usr/bin/env python3
"""Parameter gradients of one BASELINE config-3 training step (shipped model, batch 2 x 3 x 512 x 512) with the HIP
convolution weight gradient against the same step with ATen's: relative l2 difference per weight tensor."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wave_mamba_amd as wm
import bench
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = wm.WaveMamba(**bench.SHIPPED).train().to(dev)
g = torch.Generator().manual_seed(7)
lq, gt = torch.rand(2, 3, 512, 512, generator=g).to(dev), torch.rand(2, 3, 512, 512, generator=g).to(dev)
def grads(on):
    prev = wm.ops.set_train_conv_wgrad_hip(on)
    try:
        net.zero_grad(set_to_none=True)
        out = net(lq)
        loss = (out - gt).abs().mean()
        loss.backward()
        return {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
    finally:
        wm.ops.set_train_conv_wgrad_hip(prev)
a, b = grads(True), grads(False)
rows = []
for n in a:
    d = (a[n] - b[n]).norm().item(); r = b[n].norm().item()
    rows.append((d / r if r > 0 else d, n, tuple(a[n].shape)))
rows.sort(reverse=True)
print(f"{len(rows)} tensors; worst relative l2 differences (HIP weight gradient vs ATen's):")
for v, n, sh in rows[:12]:
    print(f"  {v:.3e}  {n} {sh}")
print("tensors above 1e-4:", sum(1 for r in rows if r[0] > 1e-4))
