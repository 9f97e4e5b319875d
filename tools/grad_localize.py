"""Which HIP training operator costs the gradients their accuracy?  One training step of the wf = 8 model on the GPU
path with one operator family at a time hidden from the arch (it then takes the PyTorch path for it), prediction and
worst parameter-gradient error against the float64 truth (tests/golden/train_grads_wf8_f64.npz).  Diagnostic."""
import os, sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import wave_mamba_amd as wm
from wave_mamba_amd.archs import wavemamba_arch as arch
DEV = "cuda:0"
g = np.load("tests/golden/train_grads_wf8.npz"); t = np.load("tests/golden/train_grads_wf8_f64.npz")


class Hide:
    def __init__(self, ops, hidden): self._o, self._h = ops, set(hidden)
    def __getattr__(self, k):
        if k in self._h: raise AttributeError(k)
        return getattr(self._o, k)


def err(a, b):
    d = a - b
    return max(float(d.norm() / b.norm().clamp_min(1e-300)), float(d.abs().max() / b.abs().max().clamp_min(1e-300)))


def run(hidden, label):
    arch._OpsBackend.impl = Hide(wm.ops, hidden)
    try:
        torch.manual_seed(0)
        net = wm.WaveMamba(in_chn=3, wf=8, n_l_blocks=[1, 1, 2], n_h_blocks=[1, 1, 1], ffn_scale=2.0).train()
        net.load_state_dict({k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")}, strict=False)
        net = net.to(DEV)
        pred = net(torch.from_numpy(g["lq"]).to(DEV))
        l_pix, l_fft = wm.trainer.losses(pred, torch.from_numpy(g["gt"]).to(DEV)); (l_pix + l_fft).backward()
    finally:
        arch._OpsBackend.impl = wm.ops
    tp = torch.from_numpy(t["pred"])
    rows = sorted(((err(p.grad.detach().cpu().double(), torch.from_numpy(t["t." + k])), k) for k, p in net.named_parameters()), reverse=True)
    print("%-46s pred %.2e  worst grad %.2e (%s)  >1e-4: %d  median %.2e" % (
        label, float((pred.detach().cpu().double() - tp).norm() / tp.norm()), rows[0][0], rows[0][1].replace("restoration_network.", ""),
        sum(r[0] > 1e-4 for r in rows), rows[len(rows) // 2][0]))


run([], "all HIP training operators")
for fam in (["conv2d", "conv2d_train", "conv2d_gated"], ["ss2d_core", "lfss_block_forward"], ["dwconv3x3"], ["layernorm2d"],
            ["layernorm_tok"], ["linear_nobias"], ["gram", "match_index", "attn_fold", "skff"]):
    run(fam, "without " + ", ".join(fam))
run(["conv2d", "conv2d_train", "conv2d_gated", "dwconv3x3", "layernorm2d", "layernorm_tok", "linear_nobias", "gram", "match_index",
     "attn_fold", "skff"], "only the SS2D core + DWT / IWT on HIP")
run(["conv2d", "conv2d_train", "conv2d_gated", "dwconv3x3", "layernorm2d", "layernorm_tok", "linear_nobias", "gram", "match_index",
     "attn_fold", "skff", "ss2d_core", "lfss_block_forward"], "only the drop-in scan + DWT / IWT on HIP")
