"""Worst per-parameter gradient deviations of one training step on the GPU against the reference goldens (diagnostic)."""
import json, os, sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import wave_mamba_amd as wm
DEV = "cuda:0"
if os.environ.get("WM_NO_HIP_CONV") == "1":            # experiment: PyTorch / MIOpen fp32 convolutions instead of the bf16-split kernel
    wm.ops.conv2d_supported = lambda *a, **k: False
if os.environ.get("WM_NO_HIP_CORE") == "1":            # experiment: direction glue + drop-in scan instead of the fused core
    wm.ops.ss2d_core_supported = lambda *a, **k: False
    wm.ops.lfss_block_supported = lambda *a, **k: False
gen = lambda s: torch.Generator().manual_seed(s)
meta = json.load(open("tests/golden/model_shipped_meta.json"))
torch.manual_seed(0)
net = wm.WaveMamba(in_chn=3, wf=32, n_l_blocks=[1, 2, 4], n_h_blocks=[1, 1, 2], ffn_scale=2.0).train().to(DEV)
lq = torch.rand(2, 3, 64, 64, generator=gen(1234)).to(DEV); gt = torch.rand(2, 3, 64, 64, generator=gen(4321)).to(DEV)
l_pix, l_fft = wm.trainer.losses(net(lq), gt); (l_pix + l_fft).backward()
rows = []
for k, p in net.named_parameters():
    s, a = meta["grad_fingerprint"][k]; g = p.grad.double()
    rows.append((max(abs(float(g.abs().sum()) - a), abs(float(g.sum()) - s)) / max(a, 1e-30), k, a, p.numel()))
rows.sort(reverse=True)
print("shipped fingerprints, worst 8:")
for r in rows[:8]: print("  %.3e %s abs-sum %.3e numel %d" % r)
g = np.load("tests/golden/train_grads_wf8.npz")
torch.manual_seed(0)
net = wm.WaveMamba(in_chn=3, wf=8, n_l_blocks=[1, 1, 2], n_h_blocks=[1, 1, 1], ffn_scale=2.0).train().to(DEV)
pred = net(torch.from_numpy(g["lq"]).to(DEV)); l_pix, l_fft = wm.trainer.losses(pred, torch.from_numpy(g["gt"]).to(DEV)); (l_pix + l_fft).backward()
print("pred rel", float((pred.detach().cpu() - torch.from_numpy(g["pred"])).norm() / torch.from_numpy(g["pred"]).norm()))
rows = []
for k, p in net.named_parameters():
    ref = torch.from_numpy(g["g." + k]).double(); d = p.grad.detach().cpu().double() - ref
    rows.append((max(float(d.norm() / ref.norm().clamp_min(1e-300)), float(d.abs().max() / ref.abs().max().clamp_min(1e-300))), k, float(ref.norm()), p.numel()))
rows.sort(reverse=True)
print("wf8 full tensors, worst 12:")
for r in rows[:12]: print("  %.3e %s |ref| %.3e numel %d" % r)
print("count > 1e-4:", sum(r[0] > 1e-4 for r in rows), "of", len(rows))
