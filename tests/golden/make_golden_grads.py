#!/usr/bin/env python3
"""Per-parameter gradient goldens of ONE reference training step (build container only: imports the reference arch
from /root/reference through the stub harness of make_golden.py; only tensors are written).

    python tests/golden/make_golden_grads.py        ->  tests/golden/train_grads_wf8.npz

Model: the reference WaveMamba(in_chn=3, wf=8, n_l_blocks=[1,1,2], n_h_blocks=[1,1,1], ffn_scale=2.0), weights from
torch.manual_seed(0); batch lq, gt = rand(2,3,64,64) from generators 1234 / 4321; loss of femasr_model.py:157-185
(nn.L1Loss + 0.1 * FFT-L1, losses.py:306-313); gradients of every parameter as full fp32 tensors, plus the weights,
the prediction and the two loss values.  The shipped config (1.5 M parameters) keeps float64 fingerprints in
model_shipped_meta.json; this file is the full-tensor check SURVEY.md 8c asks for.

    ->  tests/golden/train_grads_wf8_f64.npz  as well: the SAME step of the SAME reference code evaluated in float64
(`t.` keys, float64) - the truth both the reference's fp32 gradients and this build's are measured against.  A gradient
that is a short, cancelling sum (the 8 x 8 maps of the deepest block) is only defined to ~1e-4 in fp32 by ANY
implementation; the GPU test's criterion is  err(build, truth) <= max(1e-4, 2 err(reference fp32, truth))  per tensor.
The reference forces fp32 in a few places (`.float()` :123, :457-463, dtype asserts :472, :489): for this run
`Tensor.float` maps to `.double()` and `torch.float / torch.float32` name float64, nothing else is touched.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg            # noqa: E402   (stub harness + reference import)

CFG = dict(in_chn=3, wf=8, n_l_blocks=[1, 1, 2], n_h_blocks=[1, 1, 1], ffn_scale=2.0)


def main():
    torch.set_num_threads(os.cpu_count())
    arch, reg = mg.import_reference_arch()
    torch.manual_seed(0)
    net = arch.WaveMamba(**CFG).train()
    lq = torch.rand(2, 3, 64, 64, generator=mg.gen(1234))
    gt = torch.rand(2, 3, 64, 64, generator=mg.gen(4321))
    pred = net(lq)
    l_pix = F.l1_loss(pred, gt)
    pf, gf = torch.fft.rfft2(pred), torch.fft.rfft2(gt)
    l_fft = 0.1 * F.l1_loss(torch.stack([pf.real, pf.imag], -1), torch.stack([gf.real, gf.imag], -1))
    (l_pix + l_fft).backward()
    out = {"lq": mg.npy(lq), "gt": mg.npy(gt), "pred": mg.npy(pred.detach()),
           "losses": np.array([float(l_pix), float(l_fft)], dtype=np.float64)}
    n = 0
    for k, p in net.named_parameters():
        assert p.grad is not None, k
        out["w." + k] = mg.npy(p.detach())
        out["g." + k] = mg.npy(p.grad)
        n += p.numel()
    np.savez_compressed(os.path.join(HERE, "train_grads_wf8.npz"), **out)
    print(f"wrote train_grads_wf8.npz: {n} parameters in {len(out) // 2 - 2} tensors, losses {float(l_pix):.6f} {float(l_fft):.6f}")
    truth_f64(arch, net, lq, gt)
    optimizer_steps(net, lq, gt)


def optimizer_steps(net, lq, gt):
    """tests/golden/train_opt_wf8.npz: the state after optimizer_g.step() of the step above and after one more whole
    optimize_parameters() (femasr_model.py:157-185), with the optimizer built the way setup_optimizers builds it (:122-135:
    torch.optim.AdamW over every parameter with the yml's lr 5e-4, weight_decay 1e-3, betas [0.9, 0.99],
    train_wavemamba_uhdll.yml:75-79).  `net` holds the gradients of step 1.  Stored: parameters after step 1 (`p1.`),
    the two losses of step 2's forward, parameters after step 2 (`p2.`)."""
    opt = torch.optim.AdamW([p for _, p in net.named_parameters()], lr=5e-4, weight_decay=1e-3, betas=[0.9, 0.99])
    opt.step()
    out = {"p1." + k: mg.npy(p.detach()).copy() for k, p in net.named_parameters()}      # (.numpy() shares storage)
    opt.zero_grad()
    pred, l_pix, l_fft = step(net, lq, gt)
    (l_pix + l_fft).mean().backward()
    opt.step()
    out["losses2"] = np.array([float(l_pix), float(l_fft)], dtype=np.float64)
    for k, p in net.named_parameters():
        out["p2." + k] = mg.npy(p.detach())
    np.savez_compressed(os.path.join(HERE, "train_opt_wf8.npz"), **out)
    print(f"wrote train_opt_wf8.npz: losses of step 2 {float(l_pix):.6f} {float(l_fft):.6f}")


def step(net, lq, gt):
    pred = net(lq)
    l_pix = F.l1_loss(pred, gt)
    pf, gf = torch.fft.rfft2(pred), torch.fft.rfft2(gt)
    l_fft = 0.1 * F.l1_loss(torch.stack([pf.real, pf.imag], -1), torch.stack([gf.real, gf.imag], -1))
    return pred, l_pix, l_fft


def truth_f64(arch, net32, lq, gt):
    """The same weights, inputs and code in float64."""
    net = arch.WaveMamba(**CFG).train()
    net.load_state_dict(net32.state_dict())
    net = net.double()
    saved = (torch.Tensor.float, torch.float, torch.float32)
    torch.Tensor.float = lambda self, *a, **k: self.double()
    torch.float = torch.float32 = torch.float64
    try:
        pred, l_pix, l_fft = step(net, lq.double(), gt.double())
        assert pred.dtype == torch.float64
        (l_pix + l_fft).backward()
    finally:
        torch.Tensor.float, torch.float, torch.float32 = saved
    out = {"pred": mg.npy(pred.detach()), "losses": np.array([float(l_pix), float(l_fft)], dtype=np.float64)}
    worst = (0.0, None)
    for k, p in net.named_parameters():
        assert p.grad is not None and p.grad.dtype == torch.float64, k
        out["t." + k] = mg.npy(p.grad)
        g32 = net32.get_parameter(k).grad.double()
        worst = max(worst, (float((g32 - p.grad).norm() / p.grad.norm()), k))
    np.savez_compressed(os.path.join(HERE, "train_grads_wf8_f64.npz"), **out)
    print(f"wrote train_grads_wf8_f64.npz; reference fp32 vs float64 truth: worst tensor {worst[0]:.3e} ({worst[1]})")


if __name__ == "__main__":
    main()
