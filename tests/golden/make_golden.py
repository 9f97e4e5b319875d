#!/usr/bin/env python3
"""Generate golden vectors for the Wave-Mamba hot path from the reference itself.

TEST INFRASTRUCTURE - runs ONLY in the build container, where the reference tree is mounted
read-only at /root/reference.  The reference cannot travel to the GPU box, so this script imports
the reference arch file (basicsr/archs/wavemamba_arch.py) by path, runs it on seeded inputs and
commits the resulting input/output tensors as small .npz fixtures next to this file.  Nothing of
the reference's source text is stored: fixtures are data only.

How the import works: `import basicsr` fails here (cv2, timm, mamba_ssm ...
are absent), so sys.modules is pre-seeded with
  * empty package shells `basicsr`, `basicsr.utils` + the reference's own dependency-free
    `basicsr/utils/registry.py` loaded by path,
  * `timm.models.layers` exposing DropPath (identity at rate 0), to_2tuple, trunc_normal_,
  * `mamba_ssm.ops.selective_scan_interface` exposing `selective_scan_fn` / `selective_scan_ref`.

PARITY NOTE ("parity unpinned" for the scan): `mamba_ssm` is a third-party PyPI dependency,
un-vendored and un-pinned (reference requirements.txt:16).  Its arithmetic is not under
/root/reference, and the reference holds no test vectors for it.  `stub_selective_scan` below is
this build's restatement of the package's documented `selective_scan_ref` semantics (sequential
fp32 recurrence, call site wavemamba_arch.py:465-471); it is the definition of record for the scan
in this build.  Everything AROUND the scan (dwt_init, iwt_init, SS2D.forward_core glue, LFSSBlock,
the full WaveMamba network, parameter initialisers) is produced by the reference's real code.

Usage:  python tests/golden/make_golden.py            (writes tests/golden/*.npz, *.json)
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

REF_ROOT = "/root/reference"
OUT_DIR = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------------------------
# the scan stub: sequential fp32 recurrence with mamba_ssm's selective_scan_ref semantics
# --------------------------------------------------------------------------------------------
def stub_selective_scan(u, delta, A, B, C, D=None, z=None, delta_bias=None,
                        delta_softplus=False, return_last_state=False):
    in_dtype = u.dtype
    u = u.float()
    delta = delta.float()
    if delta_bias is not None:
        delta = delta + delta_bias.float().unsqueeze(-1)
    if delta_softplus:
        delta = F.softplus(delta)
    bsz, dim, seqlen = u.shape
    nstate = A.shape[1]
    # grouped, time-varying B / C: (B, G, N, L) -> (B, D, N, L)
    if B.dim() == 3:
        B = B.unsqueeze(1)
    if C.dim() == 3:
        C = C.unsqueeze(1)
    Bf = B.float().repeat_interleave(dim // B.shape[1], dim=1)
    Cf = C.float().repeat_interleave(dim // C.shape[1], dim=1)
    decay = torch.exp(delta.unsqueeze(-1) * A.float().view(1, dim, 1, nstate))        # (B,D,L,N)
    drive = (delta * u).unsqueeze(-1) * Bf.permute(0, 1, 3, 2)                        # (B,D,L,N)
    h = torch.zeros(bsz, dim, nstate, dtype=torch.float32, device=u.device)
    ys = []
    for t in range(seqlen):
        h = decay[:, :, t] * h + drive[:, :, t]
        ys.append((h * Cf[:, :, :, t]).sum(-1))
    y = torch.stack(ys, dim=2)
    if D is not None:
        y = y + u * D.float().view(1, dim, 1)
    if z is not None:
        y = y * F.silu(z.float())
    y = y.to(in_dtype)
    return (y, h) if return_last_state else y


def import_reference_arch():
    """Load /root/reference/basicsr/archs/wavemamba_arch.py with the three stub modules."""
    def shell(name, is_pkg=True):
        m = types.ModuleType(name)
        if is_pkg:
            m.__path__ = []
        sys.modules[name] = m
        return m

    shell("basicsr")
    shell("basicsr.utils")
    spec = importlib.util.spec_from_file_location(
        "basicsr.utils.registry", os.path.join(REF_ROOT, "basicsr/utils/registry.py"))
    reg = importlib.util.module_from_spec(spec)
    sys.modules["basicsr.utils.registry"] = reg
    spec.loader.exec_module(reg)

    shell("timm"); shell("timm.models")
    layers = shell("timm.models.layers", is_pkg=False)

    class DropPath(torch.nn.Module):          # rate 0 / eval -> identity
        def __init__(self, p=0.0):
            super().__init__()
            assert p == 0 or p == 0.0
        def forward(self, x):
            return x
    layers.DropPath = DropPath
    layers.to_2tuple = lambda v: (v, v) if not isinstance(v, (tuple, list)) else tuple(v)
    layers.trunc_normal_ = torch.nn.init.trunc_normal_

    shell("mamba_ssm"); shell("mamba_ssm.ops")
    ssi = shell("mamba_ssm.ops.selective_scan_interface", is_pkg=False)
    ssi.selective_scan_fn = stub_selective_scan
    ssi.selective_scan_ref = stub_selective_scan

    spec = importlib.util.spec_from_file_location(
        "ref_wavemamba_arch", os.path.join(REF_ROOT, "basicsr/archs/wavemamba_arch.py"))
    arch = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(arch)
    return arch, reg


def npy(t):
    return t.detach().cpu().contiguous().numpy()


def gen(seed):
    g = torch.Generator()
    g.manual_seed(seed)
    return g


def save(name, **arrays):
    path = os.path.join(OUT_DIR, name)
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


def main():
    torch.set_num_threads(os.cpu_count())
    arch, reg = import_reference_arch()
    assert "WaveMamba" in reg.ARCH_REGISTRY

    # ---- (i) Haar DWT / IWT (wavemamba_arch.py:97-130) -------------------------------------
    out = {}
    for tag, shape in (("a", (2, 5, 6, 10)), ("b", (1, 32, 8, 12))):
        x = torch.randn(*shape, generator=gen(1234))
        ll, hl, lh, hh = arch.dwt_init(x)
        cat = torch.cat([ll, hl, lh, hh], dim=1)
        rec = arch.iwt_init(cat)
        y = torch.randn(shape[0], 4 * shape[1], shape[2], shape[3], generator=gen(77))
        out.update({f"{tag}_x": npy(x), f"{tag}_ll": npy(ll), f"{tag}_hl": npy(hl),
                    f"{tag}_lh": npy(lh), f"{tag}_hh": npy(hh), f"{tag}_rec": npy(rec),
                    f"{tag}_iwt_in": npy(y), f"{tag}_iwt_out": npy(arch.iwt_init(y))})
    # bf16 in -> bf16 sub-bands, fp32 IWT out (reference quirk :122-123)
    xb = torch.randn(1, 4, 8, 8, generator=gen(5)).bfloat16()
    sb = arch.dwt_init(xb)
    out["bf16_x"] = npy(xb.float())
    for n, t in zip(("ll", "hl", "lh", "hh"), sb):
        assert t.dtype == torch.bfloat16
        out[f"bf16_{n}"] = npy(t.float())
    ib = arch.iwt_init(torch.cat(sb, dim=1))
    assert ib.dtype == torch.float32
    out["bf16_iwt_out"] = npy(ib)
    save("wavelet.npz", **out)

    # ---- (ii)+(iii) scan tuples harvested from SS2D.forward_core (:446-478) -----------------
    captured = []

    def spy(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
            return_last_state=False):
        y = stub_selective_scan(u, delta, A, B, C, D, z, delta_bias, delta_softplus,
                                return_last_state)
        captured.append((u, delta, A, B, C, D, delta_bias, y))
        return y

    out = {}
    for tag, (dm, ds, hh_, ww_) in {"s16": (32, 16, 8, 12), "sq16": (32, 16, 8, 8),
                                    "s32": (32, 32, 6, 10), "d8": (8, 16, 5, 7),
                                    "c32": (32, 32, 8, 12)}.items():      # d_state 32 on a map the fused core serves (W % 4 == 0)
        torch.manual_seed(0)
        ss = arch.SS2D(d_model=dm, d_state=ds, expand=2.0)
        ss.selective_scan = spy
        # make A / D / bias non-trivial so the goldens do not only see the S4D-real init
        with torch.no_grad():
            ss.A_logs.add_(0.3 * torch.randn(ss.A_logs.shape, generator=gen(11)))
            ss.Ds.add_(0.5 * torch.randn(ss.Ds.shape, generator=gen(12)))
        x = torch.randn(1 if tag != "sq16" else 2, ss.d_inner, hh_, ww_, generator=gen(1234))
        x.requires_grad_(True)
        captured.clear()
        ys = ss.forward_core(x)
        u, delta, A, B, C, D, bias, y = captured[0]
        # gradients of the scan alone, for dy = randn
        leaves = [t.detach().clone().requires_grad_(True) for t in (u, delta, A, B, C, D, bias)]
        y2 = stub_selective_scan(leaves[0], leaves[1], leaves[2], leaves[3], leaves[4], leaves[5],
                                 None, leaves[6], True, False)
        dy = torch.randn(y2.shape, generator=gen(4321))
        grads = torch.autograd.grad(y2, leaves, dy)
        # gradients of forward_core w.r.t. x and the SS2D parameters, for dys = randn
        dys = [torch.randn(t.shape, generator=gen(100 + i)) for i, t in enumerate(ys)]
        params = [ss.x_proj_weight, ss.dt_projs_weight, ss.dt_projs_bias, ss.A_logs, ss.Ds]
        gcore = torch.autograd.grad(ys, [x] + params, dys)
        out.update({
            f"{tag}_u": npy(u), f"{tag}_delta": npy(delta), f"{tag}_A": npy(A), f"{tag}_B": npy(B),
            f"{tag}_C": npy(C), f"{tag}_D": npy(D), f"{tag}_bias": npy(bias), f"{tag}_y": npy(y),
            f"{tag}_dy": npy(dy),
            f"{tag}_du": npy(grads[0]), f"{tag}_ddelta": npy(grads[1]), f"{tag}_dA": npy(grads[2]),
            f"{tag}_dB": npy(grads[3]), f"{tag}_dC": npy(grads[4]), f"{tag}_dD": npy(grads[5]),
            f"{tag}_dbias": npy(grads[6]),
            f"{tag}_core_x": npy(x), f"{tag}_x_proj_weight": npy(ss.x_proj_weight),
            f"{tag}_dt_projs_weight": npy(ss.dt_projs_weight),
            f"{tag}_dt_projs_bias": npy(ss.dt_projs_bias), f"{tag}_A_logs": npy(ss.A_logs),
            f"{tag}_Ds": npy(ss.Ds),
            **{f"{tag}_core_y{i}": npy(t) for i, t in enumerate(ys)},
            **{f"{tag}_core_dy{i}": npy(t) for i, t in enumerate(dys)},
            f"{tag}_core_dx": npy(gcore[0]), f"{tag}_core_dx_proj_weight": npy(gcore[1]),
            f"{tag}_core_ddt_projs_weight": npy(gcore[2]), f"{tag}_core_ddt_projs_bias": npy(gcore[3]),
            f"{tag}_core_dA_logs": npy(gcore[4]), f"{tag}_core_dDs": npy(gcore[5]),
        })
    # optional-argument variants of the operator surface (z gate, no D, no bias, no softplus,
    # 3-D B/C, last state) on a small synthetic tuple
    gg = gen(99)
    u = torch.randn(2, 6, 37, generator=gg); dl = 0.5 * torch.randn(2, 6, 37, generator=gg)
    A = -torch.rand(6, 5, generator=gg) * 2 - 0.1
    Bm = torch.randn(2, 5, 37, generator=gg); Cm = torch.randn(2, 5, 37, generator=gg)
    Dv = torch.randn(6, generator=gg); zz = torch.randn(2, 6, 37, generator=gg)
    bias = torch.randn(6, generator=gg) * 0.3
    y_a, last = stub_selective_scan(u, dl, A, Bm, Cm, Dv, zz, bias, True, True)
    y_b = stub_selective_scan(u, dl.abs() + 0.01, A, Bm, Cm, None, None, None, False, False)
    out.update({"opt_u": npy(u), "opt_delta": npy(dl), "opt_A": npy(A), "opt_B": npy(Bm),
                "opt_C": npy(Cm), "opt_D": npy(Dv), "opt_z": npy(zz), "opt_bias": npy(bias),
                "opt_y_full": npy(y_a), "opt_last_state": npy(last), "opt_y_plain": npy(y_b)})
    save("scan.npz", **out)

    # ---- (iv) LFSSBlock (:499-528) -----------------------------------------------------------
    torch.manual_seed(0)
    blk = arch.LFSSBlock(32, expand=2.0).eval()
    with torch.no_grad():
        blk.skip_scale.add_(0.1 * torch.randn(32, generator=gen(3)))
        blk.skip_scale2.add_(0.1 * torch.randn(32, generator=gen(4)))
    xin = torch.randn(1, 96, 32, generator=gen(1234))
    with torch.no_grad():
        yout = blk(xin, [8, 12])
    save("lfss_block.npz", x=npy(xin), y=npy(yout),
         **{"p." + k: npy(v) for k, v in blk.state_dict().items()})
    # the d_state = 32 block of BASELINE config 5, TRAINABLE: output, and the reference autograd's gradients w.r.t. the
    # input and every parameter for dy = randn (in fp32, and - the truth both are judged against, see make_golden_grads.py -
    # the same code in float64)
    torch.manual_seed(0)
    blk = arch.LFSSBlock(32, d_state=32, expand=2.0).train()
    with torch.no_grad():
        blk.skip_scale.add_(0.1 * torch.randn(32, generator=gen(3)))
        blk.skip_scale2.add_(0.1 * torch.randn(32, generator=gen(4)))
        blk.self_attention.A_logs.add_(0.3 * torch.randn(blk.self_attention.A_logs.shape, generator=gen(11)))
    xin = torch.randn(2, 16 * 12, 32, generator=gen(1234)).requires_grad_(True)
    dy = torch.randn(2, 16 * 12, 32, generator=gen(4321))
    yout = blk(xin, [16, 12])
    params = dict(blk.named_parameters())
    grads = torch.autograd.grad(yout, [xin] + list(params.values()), dy)
    blk64 = arch.LFSSBlock(32, d_state=32, expand=2.0).train()
    blk64.load_state_dict(blk.state_dict())
    blk64 = blk64.double()
    saved = (torch.Tensor.float, torch.float, torch.float32)
    torch.Tensor.float = lambda self, *a, **k: self.double()
    torch.float = torch.float32 = torch.float64
    try:
        x64 = xin.detach().double().requires_grad_(True)
        y64 = blk64(x64, [16, 12])
        g64 = torch.autograd.grad(y64, [x64] + list(blk64.parameters()), dy.double())
    finally:
        torch.Tensor.float, torch.float, torch.float32 = saved
    save("lfss_block_n32.npz", x=npy(xin), y=npy(yout), dy=npy(dy), dx=npy(grads[0]), dx_f64=npy(g64[0]), y_f64=npy(y64),
         **{"p." + k: npy(v) for k, v in blk.state_dict().items()},
         **{"g." + k: npy(g) for k, g in zip(params, grads[1:])},
         **{"t." + k: npy(g) for k, g in zip(params, g64[1:])})

    # ---- (v)+(vi) full network -----------------------------------------------------------------
    # tiny config WITH weights (does not rely on RNG-order equivalence of the re-implementation)
    torch.manual_seed(0)
    tiny = arch.WaveMamba(in_chn=3, wf=8, n_l_blocks=[1, 1, 2], n_h_blocks=[1, 1, 1],
                          ffn_scale=2.0).eval()
    xt = torch.rand(2, 3, 32, 48, generator=gen(1234))
    with torch.no_grad():
        yt = tiny(xt)
    save("model_tiny.npz", x=npy(xt), y=npy(yt),
         **{"p." + k: npy(v) for k, v in tiny.state_dict().items()})

    # shipped config (inference_wavemamba.py:71-75 == train_wavemamba_uhdll.yml:52-58), seeded init
    torch.manual_seed(0)
    net = arch.WaveMamba(in_chn=3, wf=32, n_l_blocks=[1, 2, 4], n_h_blocks=[1, 1, 2],
                         ffn_scale=2.0).eval()
    sd = net.state_dict()
    meta = {
        "n_params": int(sum(p.numel() for p in net.parameters())),
        "keys": {k: list(v.shape) for k, v in sd.items()},
        # float64 fingerprints of the seeded init (torch.manual_seed(0)); lets the GPU-box tests
        # check that the re-implementation reproduces the reference initialisation bit for bit
        "init_fingerprint": {k: [float(v.double().sum()), float(v.double().abs().sum())]
                             for k, v in sd.items()},
    }
    outs = {}
    for tag, (h, w) in {"32x64": (32, 64), "128x128": (128, 128), "256x256": (256, 256)}.items():
        xi = torch.rand(1, 3, h, w, generator=gen(1234))
        with torch.no_grad():
            yo = net(xi)
        outs[f"y_{tag}"] = npy(yo)
        meta[f"out_stats_{tag}"] = [float(yo.sum()), float(yo.mean()), float(yo.abs().max())]
        print(f"  shipped config {tag}: sum {yo.sum():.6f} mean {yo.mean():.6f} "
              f"absmax {yo.abs().max():.6f}")
    save("model_shipped.npz", **outs)
    # one training step's gradients (femasr_model.py:157-185: L1 + 0.1 * FFT-L1, losses.py:306-313)
    net.train()
    lq = torch.rand(2, 3, 64, 64, generator=gen(1234)); gt = torch.rand(2, 3, 64, 64, generator=gen(4321))
    pred = net(lq)
    l_pix = F.l1_loss(pred, gt)
    pf = torch.fft.rfft2(pred); gf = torch.fft.rfft2(gt)
    l_fft = 0.1 * F.l1_loss(torch.stack([pf.real, pf.imag], -1), torch.stack([gf.real, gf.imag], -1))
    (l_pix + l_fft).backward()
    meta["train_losses"] = [float(l_pix), float(l_fft)]
    meta["grad_fingerprint"] = {k: [float(p.grad.double().sum()), float(p.grad.double().abs().sum())]
                                for k, p in net.named_parameters()}
    # the same step of the same code in float64 (the truth the fp32 fingerprints above and the build's gradients are both
    # measured against): the reference forces fp32 at :123, :457-463 and asserts it at :472, :489, so for this run
    # Tensor.float maps to .double() and torch.float / torch.float32 name float64; nothing else is touched
    net64 = arch.WaveMamba(in_chn=3, wf=32, n_l_blocks=[1, 2, 4], n_h_blocks=[1, 1, 2], ffn_scale=2.0).train()
    net64.load_state_dict(net.state_dict())
    net64 = net64.double()
    saved = (torch.Tensor.float, torch.float, torch.float32)
    torch.Tensor.float = lambda self, *a, **k: self.double()
    torch.float = torch.float32 = torch.float64
    try:
        p64 = net64(lq.double())
        pf = torch.fft.rfft2(p64); gf = torch.fft.rfft2(gt.double())
        (F.l1_loss(p64, gt.double()) +
         0.1 * F.l1_loss(torch.stack([pf.real, pf.imag], -1), torch.stack([gf.real, gf.imag], -1))).backward()
    finally:
        torch.Tensor.float, torch.float, torch.float32 = saved
    meta["grad_fingerprint_f64"] = {k: [float(p.grad.sum()), float(p.grad.abs().sum())] for k, p in net64.named_parameters()}
    # how far the reference's own fp32 gradients are from that truth, per tensor (rel l2): the yardstick of the GPU test
    meta["grad_ref32_vs_f64"] = {k: float((net.get_parameter(k).grad.double() - p.grad).norm() / p.grad.norm().clamp_min(1e-300))
                                 for k, p in net64.named_parameters()}
    print("  reference fp32 gradients vs float64 truth: worst tensor %.3e" % max(meta["grad_ref32_vs_f64"].values()))
    with open(os.path.join(OUT_DIR, "model_shipped_meta.json"), "w") as f:
        json.dump(meta, f, indent=0)
    print("  wrote model_shipped_meta.json:", len(meta["keys"]), "keys,", meta["n_params"], "params")


if __name__ == "__main__":
    main()
