#!/usr/bin/env python3
"""Per-parameter gradient goldens of ONE reference training step (build container only: imports the reference arch
from /root/reference through the stub harness of make_golden.py; only tensors are written).

    python tests/golden/make_golden_grads.py        ->  tests/golden/train_grads_wf8.npz

Model: the reference WaveMamba(in_chn=3, wf=8, n_l_blocks=[1,1,2], n_h_blocks=[1,1,1], ffn_scale=2.0), weights from
torch.manual_seed(0); batch lq, gt = rand(2,3,64,64) from generators 1234 / 4321; loss of femasr_model.py:157-185
(nn.L1Loss + 0.1 * FFT-L1, losses.py:306-313); gradients of every parameter as full fp32 tensors, plus the weights,
the prediction and the two loss values.  The shipped config (1.5 M parameters) keeps float64 fingerprints in
model_shipped_meta.json; this file is the full-tensor check SURVEY.md 8c asks for.
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


if __name__ == "__main__":
    main()
