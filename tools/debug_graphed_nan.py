"""Debug: GraphedTrainStep replays on batches other than the captured one - which configs produce non-finite gradients?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wave_mamba_amd as wm
dev = torch.device("cuda", 0)

def trial(name, cfg, size, batch, vary, seed=11, eager_first=False):
    g = torch.Generator().manual_seed(seed)
    lq = torch.rand(3, batch, 3, size, size, generator=g).to(dev); gt = torch.rand(3, batch, 3, size, size, generator=g).to(dev)
    if eager_first:
        torch.manual_seed(0)
        ne = wm.WaveMamba(**cfg).train().to(dev); oe = wm.trainer.make_optimizer(ne, capturable=True)
        for _ in range(3):
            wm.trainer.train_step(ne, oe, lq[0], gt[0], as_float=False)
        torch.cuda.synchronize()
    torch.manual_seed(0)
    net = wm.WaveMamba(**cfg).train().to(dev); opt = wm.trainer.make_optimizer(net, capturable=True)
    step = wm.trainer.GraphedTrainStep(net, opt, lq[0], gt[0])
    out = []
    for s in range(3):
        i = s if vary else 0
        ls = step(lq[i], gt[i]); torch.cuda.synchronize()
        bad = [] if os.environ.get("NOGRAD") else [n for n, p in net.named_parameters() if not bool(torch.isfinite(p.grad).all())]
        if os.environ.get("NOGRAD") == "2":
            bad = [n for n, p in net.named_parameters() if not bool(torch.isfinite(p.detach()).all())]
        out.append((round(float(ls["l_pix"]), 6), len(bad), bad[:3]))
    print(name, "vary" if vary else "same", out, flush=True)

W8 = dict(in_chn=3, wf=8, n_l_blocks=[1, 1, 1], n_h_blocks=[1, 1, 1], ffn_scale=2.0)
W16 = dict(W8, wf=16)
W32 = dict(in_chn=3, wf=32, n_l_blocks=[1, 2, 4], n_h_blocks=[1, 1, 2], ffn_scale=2.0)
which = sys.argv[1:] or ["a", "b", "c", "d", "e"]
if "a" in which: trial("wf8 64 b2", W8, 64, 2, False)
if "b" in which: trial("wf8 64 b2", W8, 64, 2, True)
if "c" in which: trial("wf16 64 b2", W16, 64, 2, True)
if "d" in which: trial("wf16 64 b2 seed77", W16, 64, 2, True, seed=77)
if "e" in which: trial("shipped 256 b2", W32, 256, 2, True)
if "g" in which: trial("wf16 64 b2 seed77 eager-first", W16, 64, 2, True, seed=77, eager_first=True)
if "h" in which: trial("wf16 64 b2 seed77 (after g)", W16, 64, 2, True, seed=77)
if "f" in which: trial("shipped 512 b8", W32, 512, 8, True)
