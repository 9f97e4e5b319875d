"""Per-parameter gradient error of one training step on the GPU path (wf = 8 model) against the FLOAT64 evaluation of the
reference's own code (tests/golden/train_grads_wf8_f64.npz), next to the error of the reference's fp32 gradients against
the same truth (diagnostic; the test is tests/test_gpu_parity.py::test_training_step_per_parameter_gradients_on_gpu)."""
import os, sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import wave_mamba_amd as wm
DEV = "cuda:0"
if os.environ.get("WM_NO_HIP_CONV") == "1":            # experiment: PyTorch / MIOpen fp32 convolutions instead of the bf16-split kernel
    wm.ops.conv2d_supported = lambda *a, **k: False
if os.environ.get("WM_NO_HIP_CORE") == "1":            # experiment: direction glue + drop-in scan instead of the fused core
    wm.ops.ss2d_core_supported = lambda *a, **k: False
    wm.ops.lfss_block_supported = lambda *a, **k: False
g = np.load("tests/golden/train_grads_wf8.npz"); t = np.load("tests/golden/train_grads_wf8_f64.npz")
torch.manual_seed(0)
net = wm.WaveMamba(in_chn=3, wf=8, n_l_blocks=[1, 1, 2], n_h_blocks=[1, 1, 1], ffn_scale=2.0).train()
net.load_state_dict({k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")}, strict=False)
net = net.to(DEV)
pred = net(torch.from_numpy(g["lq"]).to(DEV)); l_pix, l_fft = wm.trainer.losses(pred, torch.from_numpy(g["gt"]).to(DEV)); (l_pix + l_fft).backward()
tp = torch.from_numpy(t["pred"])
print("pred rel vs truth: build %.3e  reference fp32 %.3e" % (float((pred.detach().cpu().double() - tp).norm() / tp.norm()),
                                                              float((torch.from_numpy(g["pred"]).double() - tp).norm() / tp.norm())))
def err(a, b):
    d = a - b
    return max(float(d.norm() / b.norm().clamp_min(1e-300)), float(d.abs().max() / b.abs().max().clamp_min(1e-300)))
rows = []
for k, p in net.named_parameters():
    truth = torch.from_numpy(t["t." + k])
    rows.append((err(p.grad.detach().cpu().double(), truth), err(torch.from_numpy(g["g." + k]).double(), truth), k, float(truth.norm()), p.numel()))
rows.sort(reverse=True)
print("worst 15 (build vs truth | reference fp32 vs truth):")
for r in rows[:15]: print("  %.3e | %.3e  %s |truth| %.3e numel %d" % r)
print("count build > 1e-4:", sum(r[0] > 1e-4 for r in rows), " > max(1e-4, 2 ref):", sum(r[0] > max(1e-4, 2 * r[1]) for r in rows), "of", len(rows))
