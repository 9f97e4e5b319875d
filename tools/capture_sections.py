"""Captures, in order: [0] loss forward+backward alone (L1 + FFT-L1 on a leaf), [1] L1 only, [2] FFT only, [3] network forward (train mode,
grad on), [4] network forward + backward with a plain sum loss, [5] AdamW step.  Under `rocprofv3 --hip-runtime-trace`: which capture holds
hipMemsetAsync / hipMemcpyAsync calls (= memset / memcpy graph nodes)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import wave_mamba_amd as wm
dev = torch.device("cuda", 0)
W32 = dict(in_chn=3, wf=32, n_l_blocks=[1, 2, 4], n_h_blocks=[1, 1, 2], ffn_scale=2.0)
size, batch = 256, 2
g = torch.Generator().manual_seed(11)
lq = torch.rand(batch, 3, size, size, generator=g).to(dev); gt = torch.rand(batch, 3, size, size, generator=g).to(dev)
torch.manual_seed(0)
net = wm.WaveMamba(**W32).train().to(dev); opt = wm.trainer.make_optimizer(net, capturable=True)
leaf = lq.clone().requires_grad_(True)
def sec_loss():
    leaf.grad = None; sum(wm.trainer.losses(leaf * 1.0, gt)).mean().backward()
def sec_l1():
    leaf.grad = None; F.l1_loss(leaf * 1.0, gt).backward()
def sec_fft():
    leaf.grad = None; wm.trainer.fft_l1(leaf * 1.0, gt).backward()
def sec_fwd():
    return net(lq)
def sec_fwdbwd():
    for p in net.parameters(): p.grad = None
    net(lq).sum().backward()
def sec_opt():
    opt.step()
secs = [sec_loss, sec_l1, sec_fft, sec_fwd, sec_fwdbwd, sec_opt]
side = torch.cuda.Stream(dev); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        for s in secs: s()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
keep = []
for s in secs:
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        keep.append(s())
    keep.append(gr)
torch.cuda.synchronize()
print("done")
