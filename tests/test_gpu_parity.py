"""GPU: the HIP hot path (through the C ABI) against the committed goldens and the CPU oracle.

Bars (BASELINE.json north_star / SURVEY.md 8d): Haar DWT/IWT bit-exact (pure add/sub of halves in
the reference's operation order); selective scan and anything containing it <= 1e-4 relative
(l2 and max-abs) per tensor in fp32."""
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close, rel_err
from oracle import oracle
from oracle import backend as oracle_backend
import wave_mamba_amd as wm
from wave_mamba_amd.archs import wavemamba_arch as arch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# Gradients (<= 1e-4 per tensor).  A parameter gradient that is a short, cancelling sum (the 8 x 8 maps
# of the deepest block: |g| 1e-7 .. 1e-3 against 1e-1 elsewhere) is only defined to a few 1e-5 in fp32 by ANY
# implementation, the reference's included.  The yardstick is therefore the FLOAT64 evaluation of the reference's own
# code (tests/golden/train_grads_wf8_f64.npz, model_shipped_meta.json: grad_fingerprint_f64; make_golden*.py):
#     err(build, truth) <= max(1e-4, 2 err(reference fp32, truth))   per tensor - no other allowance, and never more than
# 5e-4 in absolute terms (a tensor the reference itself gets wrong by more than 2.5e-4 would otherwise pass anything).
def truth_bar(err_ref):
    return min(max(1e-4, 2.0 * err_ref), 5e-4)


TOL = 1e-4


def gen(seed):
    g = torch.Generator()
    g.manual_seed(seed)
    return g


def cu(*ts):
    return [None if t is None else t.to(DEV) for t in ts]


# ------------------------------------------------------------------------------------------------
# Haar DWT / IWT
# ------------------------------------------------------------------------------------------------
def test_dwt_iwt_golden_bit_exact(golden):
    g = golden("wavelet")
    for tag in ("a", "b"):                      # "a": w = 5 (scalar path), "b": w = 6 (scalar path)
        outs = wm.ops.dwt_init(g[f"{tag}_x"].to(DEV))
        for name, o in zip(("ll", "hl", "lh", "hh"), outs):
            assert torch.equal(o.cpu(), g[f"{tag}_{name}"]), f"{tag}_{name}"
        assert torch.equal(wm.ops.iwt_init(g[f"{tag}_iwt_in"].to(DEV)).cpu(), g[f"{tag}_iwt_out"])


@pytest.mark.parametrize("shape", [(2, 3, 8, 16), (1, 32, 64, 96), (3, 5, 6, 10), (1, 1, 2, 2),
                                   (2, 7, 34, 40), (1, 4, 16, 8200)])
def test_dwt_iwt_vs_oracle(shape):
    x = torch.randn(*shape, generator=gen(1))
    got = wm.ops.dwt_init(x.to(DEV))
    ref = oracle.dwt_raw(x)
    for a, b in zip(got, ref):
        assert torch.equal(a.cpu(), b)
    y = torch.randn(shape[0], 4 * shape[1], shape[2], shape[3], generator=gen(2))
    ref_i = oracle.iwt_raw(y)
    assert torch.equal(wm.ops.iwt_init(y.to(DEV)).cpu(), ref_i)
    C = shape[1]
    assert torch.equal(wm.ops.iwt_init_pair(y[:, :C].contiguous().to(DEV), y[:, C:].contiguous().to(DEV)).cpu(),
                       ref_i)


def test_dwt_bf16_matches_reference_rounding(golden):
    g = golden("wavelet")
    outs = wm.ops.dwt_init(g["bf16_x"].to(DEV).bfloat16())
    for name, o in zip(("ll", "hl", "lh", "hh"), outs):
        assert o.dtype == torch.bfloat16
        assert torch.equal(o.float().cpu(), g[f"bf16_{name}"]), name
    rec = wm.ops.iwt_init(torch.cat(outs, 1))
    assert rec.dtype == torch.float32                       # reference quirk: IWT is always fp32
    assert torch.equal(rec.cpu(), g["bf16_iwt_out"])
    # vector path too (w % 4 == 0)
    x = torch.randn(2, 4, 16, 64, generator=gen(3)).bfloat16()
    outs = wm.ops.dwt_init(x.to(DEV))
    a, b = x[:, :, 0::2, :] / 2, x[:, :, 1::2, :] / 2      # eager bf16 arithmetic = the reference's
    x1, x2, x3, x4 = a[..., 0::2], b[..., 0::2], a[..., 1::2], b[..., 1::2]
    for got, want in zip(outs, (x1 + x2 + x3 + x4, -x1 - x2 + x3 + x4, -x1 + x2 - x3 + x4, x1 - x2 - x3 + x4)):
        assert torch.equal(got.cpu(), want)


def test_dwt_errors():
    with pytest.raises(RuntimeError):
        wm.ops.dwt_init(torch.zeros(1, 1, 5, 4, device=DEV))       # odd H: reference raises too
    xc = torch.randn(1, 3, 6, 8, generator=gen(9))                 # a CPU tensor runs the CPU twin (cpu_twin.py): same bits
    for got, want in zip(wm.ops.dwt_init(xc), wm.ops.dwt_init(xc.to(DEV))):
        assert got.device.type == "cpu" and torch.equal(got, want.cpu())
    with pytest.raises(RuntimeError):                              # mixed devices: neither implementation takes them
        wm.ops.iwt_init_pair(torch.zeros(1, 1, 4, 4), torch.zeros(1, 3, 4, 4, device=DEV))
    assert all(o.numel() == 0 for o in wm.ops.dwt_init(torch.zeros(0, 3, 4, 4, device=DEV)))


def test_dwt_iwt_backward_is_the_adjoint():
    x = torch.randn(2, 6, 12, 20, generator=gen(4)).to(DEV).requires_grad_(True)
    outs = wm.ops.dwt_init(x)
    gs = [torch.randn(o.shape, generator=gen(10 + i)).to(DEV) for i, o in enumerate(outs)]
    (dx,) = torch.autograd.grad(outs, x, gs)
    assert torch.equal(dx.cpu(), oracle.iwt_raw(torch.cat([g.cpu() for g in gs], 1)))
    y = torch.randn(2, 24, 6, 10, generator=gen(5)).to(DEV).requires_grad_(True)
    out = wm.ops.iwt_init(y)
    g = torch.randn(out.shape, generator=gen(6)).to(DEV)
    (dy,) = torch.autograd.grad(out, y, g)
    assert torch.equal(dy.cpu(), torch.cat(oracle.dwt_raw(g.cpu()), 1))
    yl, yh = y.detach()[:, :6].contiguous().requires_grad_(True), y.detach()[:, 6:].contiguous().requires_grad_(True)
    dl, dh = torch.autograd.grad(wm.ops.iwt_init_pair(yl, yh), (yl, yh), g)
    assert torch.equal(torch.cat([dl, dh], 1), dy)


def test_dwt_full_size_bit_exact_and_properties():
    """BASELINE config 4 at its full size (UHDLOL4K: batch 4 x 32 x 2160 x 4096 fp32 = 4.5 GB, 3 DWT + 3 IWT levels):
    every sub-band of every level and every reconstruction BIT-EXACT against the CPU oracle over the whole tensor (the
    oracle runs one batch item at a time to bound host memory; the GPU calls are the full-size ones), plus the
    size-independent properties (energy, round trip, linearity)."""
    x_cpu = torch.randn(4, 32, 2160, 4096, generator=gen(7))
    x = x_cpu.to(DEV)
    cur, cur_cpu, pyramid = x, x_cpu, []
    for level in range(3):
        ll, hl, lh, hh = wm.ops.dwt_init(cur)
        want = [oracle.dwt_raw(cur_cpu[b:b + 1]) for b in range(cur_cpu.shape[0])]
        for i, (name, got) in enumerate(zip(("ll", "hl", "lh", "hh"), (ll, hl, lh, hh))):
            for b in range(cur_cpu.shape[0]):
                assert torch.equal(got[b:b + 1].cpu(), want[b][i]), f"level {level + 1} {name} batch {b}"
        e_in = cur.double().pow(2).sum()
        e_out = sum(t.double().pow(2).sum() for t in (ll, hl, lh, hh))
        assert abs(float(e_out / e_in) - 1.0) < 1e-6          # orthogonal transform
        pyramid.append((hl, lh, hh))
        cur, cur_cpu = ll, torch.cat([w[0] for w in want], 0)
        del want
    del cur_cpu
    for hl, lh, hh in reversed(pyramid):
        cat = torch.cat([cur, hl, lh, hh], 1)
        cur = wm.ops.iwt_init(cat)
        for b in range(cat.shape[0]):
            assert torch.equal(cur[b:b + 1].cpu(), oracle.iwt_raw(cat[b:b + 1].cpu())), f"IWT to {tuple(cur.shape)} batch {b}"
        del cat
    assert float((cur - x).abs().max()) < 5e-6
    # linearity: DWT(a + 2b) == DWT(a) + 2 DWT(b) up to rounding
    a, b = x[:, :4, :256, :512].contiguous(), torch.randn(1, 4, 256, 512, generator=gen(8)).to(DEV)
    for p, q, r in zip(wm.ops.dwt_init(a + 2 * b), wm.ops.dwt_init(a), wm.ops.dwt_init(b)):
        assert float((p - (q + 2 * r)).abs().max()) < 1e-5


# ------------------------------------------------------------------------------------------------
# selective scan forward
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["s16", "sq16", "s32", "d8"])
def test_scan_golden(golden, tag):
    g = golden("scan")
    y = wm.ops.selective_scan_fn(*cu(g[f"{tag}_u"], g[f"{tag}_delta"], g[f"{tag}_A"], g[f"{tag}_B"],
                                     g[f"{tag}_C"], g[f"{tag}_D"]), None, g[f"{tag}_bias"].to(DEV), True)
    assert y.dtype == torch.float32
    assert_close(y, g[f"{tag}_y"], TOL, f"{tag} y")


def test_scan_optional_arguments(golden):
    g = golden("scan")
    y, last = wm.ops.selective_scan_fn(*cu(g["opt_u"], g["opt_delta"], g["opt_A"], g["opt_B"], g["opt_C"],
                                           g["opt_D"], g["opt_z"], g["opt_bias"]), True, True)
    assert_close(y, g["opt_y_full"], TOL, "z-gated y")
    assert_close(last, g["opt_last_state"], TOL, "last state")
    y2 = wm.ops.selective_scan_fn(*cu(g["opt_u"], g["opt_delta"].abs() + 0.01, g["opt_A"], g["opt_B"], g["opt_C"]))
    assert_close(y2, g["opt_y_plain"], TOL, "plain y")


def random_scan_case(batch, dim, L, N, G, seed, softplus_bias=-3.0):
    gg = gen(seed)
    u = torch.randn(batch, dim, L, generator=gg)
    delta = 0.5 * torch.randn(batch, dim, L, generator=gg)
    A = -torch.exp(torch.log(torch.arange(1, N + 1, dtype=torch.float32)).repeat(dim, 1)
                   + 0.2 * torch.randn(dim, N, generator=gg))
    Bm = torch.randn(batch, G, N, L, generator=gg)
    Cm = torch.randn(batch, G, N, L, generator=gg)
    D = torch.randn(dim, generator=gg)
    bias = 0.5 * torch.randn(dim, generator=gg) + softplus_bias
    return u, delta, A, Bm, Cm, D, bias


@pytest.mark.parametrize("batch,dim,L,N,G", [
    (1, 256, 16384, 16, 4),      # BASELINE config 1, level 1 (128 x 128): multi-chunk
    (2, 256, 4096, 16, 4),
    (1, 256, 1024, 16, 4),
    (1, 64, 5000, 16, 1),        # one group of 64
    (1, 256, 3000, 32, 4),       # d_state 32 (BASELINE config 5 flavour)
    (2, 96, 777, 16, 1),         # 96 channels per group (wf = 48): 2 waves per group, ragged; L % 4 != 0
    (1, 40, 130, 7, 4),          # 10 channels per group, N = 7 (padded), scalar path
    (3, 8, 1, 4, 2),             # L = 1
    (1, 128, 66000, 16, 2),      # many chunks, L % 16 != 0
])
def test_scan_vs_oracle(batch, dim, L, N, G):
    case = random_scan_case(batch, dim, L, N, G, seed=batch * 1000 + L)
    y, last = wm.ops.selective_scan_fn(*cu(*case[:6]), None, case[6].to(DEV), True, True)
    yr, lr = oracle.selscan_fwd_raw(*case[:6], None, case[6], True, True)
    assert_close(y, yr, TOL, "y")
    assert_close(last, lr, TOL, "last_state")


def test_scan_stress_inputs():
    # large positive delta (softplus threshold branch), tiny delta, zero u: no NaN, matches oracle
    u, delta, A, Bm, Cm, D, bias = random_scan_case(1, 64, 2048, 16, 1, seed=5, softplus_bias=0.0)
    delta[:, :, 100:110] = 30.0          # softplus(x) = x above 20; exp(dt*A) underflows to 0
    delta[:, :, 500:600] = -40.0         # dt ~ 4e-18
    u[:, :, 900:1000] = 0.0
    y = wm.ops.selective_scan_fn(*cu(u, delta, A, Bm, Cm, D), None, bias.to(DEV), True)
    assert torch.isfinite(y).all()
    assert_close(y, oracle.selscan_fwd_raw(u, delta, A, Bm, Cm, D, None, bias, True), TOL, "stress y")


def test_scan_errors():
    u, delta, A, Bm, Cm, D, bias = cu(*random_scan_case(1, 8, 16, 4, 2, seed=1))
    with pytest.raises(RuntimeError):
        wm.ops.selective_scan_fn(u, delta[:, :, :8], A, Bm, Cm)
    with pytest.raises(RuntimeError):
        wm.ops.selective_scan_fn(u, delta, A, Bm[:, :, :3], Cm)
    with pytest.raises(RuntimeError):                              # mixed devices: refused (all-CPU inputs run the CPU twin)
        wm.ops.selective_scan_fn(u.cpu(), delta, A, Bm, Cm)
    assert_close(wm.ops.selective_scan_fn(u.cpu(), delta.cpu(), A.cpu(), Bm.cpu(), Cm.cpu()),
                 wm.ops.selective_scan_fn(u, delta, A, Bm, Cm).cpu(), TOL, "CPU twin vs HIP")
    with pytest.raises(NotImplementedError):
        wm.ops.selective_scan_fn(u, delta, A, A, A)


def test_scan_uhd_level1_properties():
    """BASELINE config 2, level-1 scan at full size (B=1, KD=256, L=2,088,960): the oracle on all
    channels would take minutes, so check (a) 8 channels of every group against the oracle and
    (b) linearity in u on the full tensor."""
    L, dim, N, G = 1088 * 1920, 256, 16, 4
    gg = torch.Generator(device=DEV)
    gg.manual_seed(11)
    u = torch.randn(1, dim, L, generator=gg, device=DEV)
    delta = 0.5 * torch.randn(1, dim, L, generator=gg, device=DEV)
    Bm = torch.randn(1, G, N, L, generator=gg, device=DEV)
    Cm = torch.randn(1, G, N, L, generator=gg, device=DEV)
    A = -torch.arange(1, N + 1, dtype=torch.float32, device=DEV).repeat(dim, 1) * \
        torch.exp(0.2 * torch.randn(dim, N, generator=gg, device=DEV))
    D = torch.randn(dim, generator=gg, device=DEV)
    bias = 0.5 * torch.randn(dim, generator=gg, device=DEV) - 4.0
    y = wm.ops.selective_scan_fn(u, delta, A, Bm, Cm, D, None, bias, True)
    assert torch.isfinite(y).all()
    sel = torch.cat([torch.arange(g * 64 + 5, g * 64 + 61, 8) for g in range(G)])       # 7 per group
    sel = torch.cat([sel, torch.tensor([0, 63, 64, 255])]).sort().values
    # keep the group structure: pick the same number of channels in every group
    per_group = [sel[(sel >= g * 64) & (sel < (g + 1) * 64)] for g in range(G)]
    k = min(len(p) for p in per_group)
    sel = torch.cat([p[:k] for p in per_group])
    yr = oracle.selscan_fwd_raw(u[:, sel].cpu(), delta[:, sel].cpu(), A[sel].cpu(), Bm.cpu(), Cm.cpu(),
                                D[sel].cpu(), None, bias[sel].cpu(), True)
    assert_close(y[:, sel], yr, TOL, "UHD level-1 subset")
    u2 = torch.randn(1, dim, L, generator=gg, device=DEV)
    y2 = wm.ops.selective_scan_fn(u2, delta, A, Bm, Cm, D, None, bias, True)
    y12 = wm.ops.selective_scan_fn(u + 2 * u2, delta, A, Bm, Cm, D, None, bias, True)
    assert_close(y12, y + 2 * y2, 1e-5, "linearity in u")


# ------------------------------------------------------------------------------------------------
# selective scan backward (autograd of selective_scan_fn; training path, femasr_model.py:181)
# ------------------------------------------------------------------------------------------------
GRAD_NAMES = ("du", "ddelta", "dA", "dB", "dC", "dD", "dbias")


def hip_scan_grads(u, delta, A, Bm, Cm, D, bias, dy):
    leaves = [t.to(DEV).requires_grad_(True) for t in (u, delta, A, Bm, Cm, D, bias)]
    y = wm.ops.selective_scan_fn(leaves[0], leaves[1], leaves[2], leaves[3], leaves[4], leaves[5], None,
                                 leaves[6], True)
    return y, torch.autograd.grad(y, leaves, dy.to(DEV))


@pytest.mark.parametrize("tag", ["s16", "sq16", "s32", "d8"])
def test_scan_backward_golden(golden, tag):
    g = golden("scan")           # gradients from autograd through the sequential reference recurrence
    y, grads = hip_scan_grads(*[g[f"{tag}_{k}"] for k in ("u", "delta", "A", "B", "C", "D", "bias", "dy")])
    assert_close(y, g[f"{tag}_y"], TOL, f"{tag} y")
    for name, got in zip(GRAD_NAMES, grads):
        assert_close(got, g[f"{tag}_{name}"], TOL, f"{tag} {name}")


@pytest.mark.parametrize("batch,dim,L,N,G", [
    (2, 256, 4096, 16, 4),       # BASELINE config 3 flavour (512-crop level 3): many chunks, batch
    (1, 64, 1000, 16, 1),        # L % 16 != 0
    (1, 256, 300, 32, 4),        # d_state 32
    (2, 96, 77, 16, 1),          # 96 channels per group: two waves per group (atomic dB/dC), L % 4 != 0
    (1, 40, 50, 7, 4),           # 10 channels per group, N = 7
    (1, 8, 1, 4, 2),             # L = 1
])
def test_scan_backward_vs_oracle(batch, dim, L, N, G):
    """du / ddelta / dB / dC against the CPU oracle at 1e-4.  dA / dD / dbias are sums over batch and L of products of
    both signs: they are judged like every reduced gradient - against a float64 evaluation of the same recurrence
    (autograd through the sequential definition, the oracle's fp32 result measured against it too)."""
    case = random_scan_case(batch, dim, L, N, G, seed=7 + L)
    dy = torch.randn(batch, dim, L, generator=gen(L))
    _, grads = hip_scan_grads(*case, dy)
    want = oracle.selscan_bwd_raw(*case, dy, True)
    truth = scan_grads_f64(*case, dy)
    for name, got, ref in zip(GRAD_NAMES, grads, want):
        if name in ("dA", "dD", "dbias") and float(truth[name].abs().max()) == 0.0:      # L = 1: dA is exactly zero
            assert float(got.abs().max()) <= 1e-6, f"{name}: expected zeros, max abs {float(got.abs().max()):.3e}"
        elif name in ("dA", "dD", "dbias"):
            e_ref = max(rel_err(ref.double(), truth[name]))
            e_got = max(rel_err(got.cpu().double(), truth[name]))
            assert e_got <= truth_bar(e_ref), f"{name}: {e_got:.3e} vs float64 truth (oracle fp32: {e_ref:.3e})"
        else:
            assert_close(got, ref, TOL, f"{name}")


def scan_grads_f64(u, delta, A, Bm, Cm, D, bias, dy):
    """The selective-scan definition in float64 with autograd: truth for the reduced gradients."""
    leaves = [t.double().requires_grad_(True) for t in (u, delta, A, Bm, Cm, D, bias)]
    u_, dl, A_, B_, C_, D_, b_ = leaves
    batch, dim, L = u_.shape
    N, G = A_.shape[1], B_.shape[1]
    dt = F.softplus(dl + b_.view(1, dim, 1))
    Bf = B_.repeat_interleave(dim // G, dim=1)          # (b, dim, N, L)
    Cf = C_.repeat_interleave(dim // G, dim=1)
    h = torch.zeros(batch, dim, N, dtype=torch.float64)
    ys = []
    # (unbind, not indexing: the backward of `t[..., i]` allocates a full-size zero tensor per step)
    for dt_t, du_t, B_t, C_t in zip(dt.unbind(2), (dt * u_).unbind(2), Bf.unbind(3), Cf.unbind(3)):
        h = torch.exp(dt_t[..., None] * A_.view(1, dim, N)) * h + du_t[..., None] * B_t
        ys.append((h * C_t).sum(-1))
    y = torch.stack(ys, 2) + u_ * D_.view(1, dim, 1)
    g = torch.autograd.grad(y, leaves, dy.double())
    return dict(zip(GRAD_NAMES, g))


def test_training_step_on_gpu_matches_reference(golden):
    """One optimize_parameters() of the reference trainer on the HIP path (module glue + HIP scan fwd/bwd +
    HIP DWT/IWT fwd/bwd) against the reference's gradient fingerprints (tests/golden/model_shipped_meta.json)."""
    import json, os
    from conftest import GOLDEN
    meta = json.load(open(os.path.join(GOLDEN, "model_shipped_meta.json")))
    torch.manual_seed(0)
    net = wm.WaveMamba(in_chn=3, wf=32, n_l_blocks=[1, 2, 4], n_h_blocks=[1, 1, 2], ffn_scale=2.0).train().to(DEV)
    lq = torch.rand(2, 3, 64, 64, generator=gen(1234)).to(DEV)
    gt = torch.rand(2, 3, 64, 64, generator=gen(4321)).to(DEV)
    l_pix, l_fft = wm.trainer.losses(net(lq), gt)
    (l_pix + l_fft).backward()
    assert abs(float(l_pix.detach()) - meta["train_losses"][0]) < 1e-5
    # (sum, abs-sum) fingerprints only at 1.5 M parameters (the full tensors are checked on the wf = 8 model below):
    # the build's and the reference's fp32 fingerprints against the float64 ones
    fp = lambda a, b: max(abs(a[0] - b[0]), abs(a[1] - b[1])) / max(b[1], 1e-30)
    bad = []
    for k, p in net.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
        g = p.grad.double()
        truth = meta["grad_fingerprint_f64"][k]
        e_got = fp((float(g.sum()), float(g.abs().sum())), truth)
        e_ref = max(fp(meta["grad_fingerprint"][k], truth), meta["grad_ref32_vs_f64"][k])
        if e_got > truth_bar(e_ref):
            bad.append((e_got, e_ref, k))
    assert not bad, "gradient fingerprints off the float64 truth (build, reference fp32, name): " + \
        "; ".join("%.3e %.3e %s" % b for b in sorted(bad, reverse=True)[:5])


def test_training_step_per_parameter_gradients_on_gpu():
    """One reference training step (femasr_model.py:157-185) on the HIP training path - SS2D core forward / backward,
    DWT / IWT, depth-wise conv, LayerNorms, convolutions' input gradients in HIP, the rest PyTorch autograd: every
    parameter's gradient TENSOR against the float64 evaluation of the reference's code, judged against what the
    reference's own fp32 autograd (tests/golden/train_grads_wf8.npz) achieves on the same tensor."""
    import numpy as np, os
    from conftest import GOLDEN
    from test_arch_cpu import grad_golden_case
    net, g = grad_golden_case()
    t = np.load(os.path.join(GOLDEN, "train_grads_wf8_f64.npz"))
    net = net.to(DEV)
    pred = net(torch.from_numpy(g["lq"]).to(DEV))
    l_pix, l_fft = wm.trainer.losses(pred, torch.from_numpy(g["gt"]).to(DEV))
    (l_pix + l_fft).backward()
    assert abs(float(l_pix.detach()) - g["losses"][0]) < 1e-6
    assert_close(pred.detach(), torch.from_numpy(g["pred"]), 1e-5, "prediction")
    bad, worst = [], (0.0, 0.0, None)
    for k, p in net.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
        truth = torch.from_numpy(t["t." + k])
        e_got = max(rel_err(p.grad.detach().cpu().double(), truth))
        e_ref = max(rel_err(torch.from_numpy(g["g." + k]).double(), truth))
        worst = max(worst, (e_got, e_ref, k))
        if e_got > truth_bar(e_ref):
            bad.append((e_got, e_ref, k))
    print("worst per-parameter gradient error vs float64 truth: %.3e (reference fp32 %.3e) %s" % worst)
    assert not bad, "gradients off the float64 truth (build, reference fp32, name): " + \
        "; ".join("%.3e %.3e %s" % b for b in sorted(bad, reverse=True)[:8])


# ------------------------------------------------------------------------------------------------
# fused SS2D four-direction core (reference SS2D.forward_core, :446-478)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["s16", "sq16", "c32", "d8"])       # c32: d_state 32 on an 8 x 12 map (W % 4 == 0)
def test_ss2d_core_golden(golden, tag):
    g = golden("scan")          # captured from the reference's real forward_core
    args = cu(g[f"{tag}_core_x"], g[f"{tag}_x_proj_weight"], g[f"{tag}_dt_projs_weight"],
              g[f"{tag}_dt_projs_bias"], g[f"{tag}_A_logs"], g[f"{tag}_Ds"])
    ys = wm.ops.ss2d_core(*args)
    for i, y in enumerate(ys):
        assert_close(y, g[f"{tag}_core_y{i}"], TOL, f"{tag} core y{i}")
    ysum = wm.ops.ss2d_core(*args, merged=True)
    assert_close(ysum, sum(g[f"{tag}_core_y{i}"] for i in range(4)), TOL, f"{tag} merged")


def random_core_case(B, D, H, W, N, R, seed):
    gg = gen(seed)
    x = torch.randn(B, D, H, W, generator=gg)
    Wx = torch.randn(4, R + 2 * N, D, generator=gg) / D ** 0.5
    Wdt = torch.randn(4, D, R, generator=gg) * R ** -0.5
    bias = torch.randn(4, D, generator=gg) * 0.5 - 3.0
    A_logs = torch.log(torch.arange(1, N + 1, dtype=torch.float32)).repeat(4 * D, 1) + 0.2 * torch.randn(4 * D, N, generator=gg)
    Ds = torch.randn(4 * D, generator=gg)
    return x, Wx, Wdt, bias, A_logs, Ds


@pytest.mark.parametrize("B,D,H,W,N,R", [
    (1, 64, 32, 32, 16, 2),      # BASELINE config 1, level 3
    (1, 64, 64, 64, 16, 2),      # level 2: several chunks / segments
    (2, 64, 24, 40, 16, 2),      # batch 2, W < 64 (ragged column tile)
    (1, 64, 40, 136, 16, 2),     # 3 column tiles, last ragged
    (1, 16, 10, 14, 16, 1),      # wf = 8: d_inner 16, dt_rank 1
    (1, 48, 9, 7, 8, 3),         # odd sizes (L % 4 != 0 -> scalar paths), N = 8, R = 3
    (1, 64, 128, 128, 16, 2),    # config 1 level 1: many chunks, several row segments
    (1, 16, 16, 2048, 16, 2),    # 512 row chunks (one-level carry) next to 2048 column chunks (two-level): separate carry batches
    (1, 16, 8192, 16, 16, 2),    # the other way round: 2048 row chunks, few column chunks
    (1, 64, 48, 40, 32, 2),      # d_state 32 (BASELINE config 5's block): 8-wave workgroups, 8-column tiles, ragged W
    (2, 64, 33, 72, 32, 4),      # d_state 32, dt_rank 4, odd H, batch 2
    (1, 64, 50, 52, 16, 4),      # dt_rank 4 at d_state 16; W % 16 != 0, H % 16 != 0
    (1, 32, 272, 480, 16, 1),    # UHD level-3 map at wf = 16: column segments
    (1, 64, 9, 7, 32, 2),        # d_state 32 on an odd map (element-wise tile accesses; the first-generation kernels
                                 # that served odd widths until round 3 stopped at d_state 16)
    (2, 64, 33, 71, 16, 2),      # odd width, several column tiles and row chunks, batch 2 (odd plane offsets)
    (1, 40, 37, 130, 32, 3),     # W % 4 == 2, d_state 32, dt_rank 3, D < 64
    (1, 64, 65, 33, 16, 2),      # L odd, several row chunks: reversed row tiles start before the plane
    (1, 8, 1, 1, 16, 1),         # a single position
    (1, 64, 1, 37, 16, 2),       # a single row / a single column
    (1, 64, 37, 1, 16, 2),
])
def test_ss2d_core_vs_oracle(B, D, H, W, N, R):
    case = random_core_case(B, D, H, W, N, R, seed=H * 100 + W)
    want = oracle.ss2d_core_raw(*case)
    got = wm.ops.ss2d_core(*cu(*case))
    for i, (a, b) in enumerate(zip(got, want)):
        assert_close(a, b, TOL, f"core y{i}")
    assert_close(wm.ops.ss2d_core(*cu(*case), merged=True), sum(want), TOL, "merged")


@pytest.mark.parametrize("B,Cin,H,W,r,Cout", [
    (1, 3, 64, 96, 2, 32), (2, 3, 64, 96, 4, 32), (1, 3, 64, 96, 8, 32), (1, 3, 272, 520, 2, 32), (1, 3, 40, 1048, 8, 32),
    (2, 3, 16, 24, 8, 16), (1, 3, 32, 48, 4, 48), (1, 4, 24, 40, 2, 64), (1, 1, 8, 8, 8, 16), (1, 3, 2160, 3840, 8, 32),
])
def test_patchify_conv_vs_float64(B, Cin, H, W, r, Cout):
    """wm_patchify_conv_fwd = nn.PixelUnshuffle(r) + 1x1 nn.Conv2d (reference :1014-1025) against the float64 composition of the two
    PyTorch ops; the fp32 composition's own error is printed beside it."""
    g = gen(H + W + r)
    img = torch.rand(B, Cin, H, W, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin * r * r, 1, 1, generator=g) / (Cin * r * r) ** 0.5).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV)
    assert wm.ops.patchify_conv_supported(img, w, r)
    got = wm.ops.patchify_conv(img, w, b, r)
    want = F.conv2d(F.pixel_unshuffle(img.double(), r), w.double(), b.double())
    assert_close(got.double(), want, 2e-6, f"patchify r={r}")
    nob = wm.ops.patchify_conv(img, w, None, r)
    assert_close(nob.double(), want - b.double().view(1, -1, 1, 1), 2e-6, f"patchify r={r} no bias")


def test_patchify_conv_in_network(golden):
    """The UNet takes the fused kernel for ps_down1..3 in inference: same output as with the two modules (PixelUnshuffle copy +
    1x1 matrix-core convolution) within the convolutions' own accuracy."""
    torch.manual_seed(0)
    net = wm.WaveMamba(in_chn=3, wf=32, n_l_blocks=[1, 1, 1], n_h_blocks=[1, 1, 1], ffn_scale=2.0).eval().to(DEV)
    img = torch.rand(1, 3, 128, 192, generator=gen(3)).to(DEV)
    with torch.no_grad():
        a = net(img)
        orig = wm.ops.patchify_conv_supported
        try:
            wm.ops.patchify_conv_supported = lambda *args: False
            b = net(img)
        finally:
            wm.ops.patchify_conv_supported = orig
    assert_close(a, b, 2e-5, "network with / without the fused patch embedding")


def test_ss2d_core_unaligned_planes():
    """x at a 4-byte-aligned address that is not 16-byte aligned (a view into a larger buffer): the core takes its
    element-wise tile accesses instead of refusing the call; same results as from an aligned copy."""
    case = random_core_case(1, 64, 24, 40, 16, 2, seed=5)
    dev = cu(*case)
    big = torch.empty(dev[0].numel() + 1, device=DEV)
    xv = big[1:].view_as(dev[0]); xv.copy_(dev[0])
    assert xv.data_ptr() % 16 == 4 and xv.is_contiguous()
    want = wm.ops.ss2d_core(*dev)
    got = wm.ops.ss2d_core(xv, *dev[1:])
    for i, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a, b), f"y{i}: element-wise and 16-byte tile accesses differ"


def core_vs_unfused_and_oracle_subset(H, W, N, seed, channels, what, full_direction=None):
    """Full-size check of the fused core (the oracle on all channels would take minutes):
    (a) all four outputs against the direction glue + the drop-in selective_scan_fn (oracle-checked on its own);
    (b) `channels` of every direction against the CPU oracle's scan, with the operands of those channels formed the
        way the reference forms them (:451-455) in float64 on the GPU."""
    D, R, K = 64, 2, 4
    case = cu(*random_core_case(1, D, H, W, N, R, seed=seed))
    x, Wx, Wdt, bias, A_logs, Ds = case
    L = H * W
    ss = arch.SS2D(d_model=32, d_state=N, expand=2.0).to(DEV)
    with torch.no_grad():
        ss.x_proj_weight.copy_(Wx); ss.dt_projs_weight.copy_(Wdt); ss.dt_projs_bias.copy_(bias)
        ss.A_logs.copy_(A_logs); ss.Ds.copy_(Ds)
        fused = wm.ops.ss2d_core(*case)
        ss._fused_ok = lambda _x: False                     # force the unfused path
        unfused = ss.forward_core(x)
    for i, (a, b) in enumerate(zip(fused, unfused)):
        assert_close(a, b, TOL, f"{what} core y{i} vs unfused")
    del unfused
    # reference return order (:478): y_row_fwd (k=0), y_row_rev (k=2), y_col_fwd (k=1), y_col_rev (k=3)
    order = {0: 0, 2: 1, 1: 2, 3: 3}
    for k in range(K):
        chs = list(range(D)) if k == full_direction else channels    # (4 of 64 channels per direction were checked)
        sel = torch.tensor(chs, device=DEV)
        xs = x.reshape(1, D, L) if k % 2 == 0 else x.transpose(2, 3).reshape(1, D, L)      # l = h W + w | l = w H + h
        if k >= 2:
            xs = xs.flip(-1)
        x_dbl = torch.einsum("bdl,cd->bcl", xs.double(), Wx[k].double())
        dts_r, Bs, Cs = torch.split(x_dbl, [R, N, N], dim=1)
        delta = torch.einsum("brl,dr->bdl", dts_r, Wdt[k, sel].double()).float()
        kd = k * D + sel
        y = oracle.selscan_fwd_raw(xs[:, sel].cpu(), delta.cpu(), -torch.exp(A_logs[kd]).cpu(), Bs.float().cpu().unsqueeze(1),
                                   Cs.float().cpu().unsqueeze(1), Ds[kd].cpu(), None, bias[k, sel].cpu(), True)
        if k >= 2:
            y = y.flip(-1)
        if k % 2 == 1:                                                                     # back to row-major l
            y = y.reshape(1, len(chs), W, H).transpose(2, 3).reshape(1, len(chs), L)
        assert_close(fused[order[k]][:, sel], y, TOL, f"{what} direction {k} vs oracle on {len(chs)} channels")


# ------------------------------------------------------------------------------------------------
# fused SS2D core OUTSIDE the default-init distribution (VERDICT r5 "missing" 4): trained-like / adversarial parameters
# ------------------------------------------------------------------------------------------------
def ood_core_case(B, D, H, W, N, R, seed, kind):
    """SS2D.forward_core operands no default init produces (the fused core has its own log2-unit softplus without F.softplus's
    threshold branch, a bf16-split x_proj and prepared A - reference edge: `delta_softplus=True` with threshold 20,
    wavemamba_arch.py:465-471):
      trained   dt_projs_bias ~ U[-8, 4], A_logs ~ U[-3, 6] (non-monotone in n: A from -0.05 to -400), Ds ~ N(0, 1), projections x3 / x4
      dtpush    + bands of rows where x is 40x larger: the dt pre-activation runs beyond +20 (softplus's linear branch) and below
                -40 (dt ~ 1e-18) over whole runs of positions, per channel sign
      small / large   x scaled by 1e-3 / 1e+2
      zeroplane one channel plane of x all zero (and one whole batch image where B > 1)
      zeroD     Ds = 0
      zerox     x = 0 everywhere (y must be exactly 0)"""
    gg = gen(seed)
    x = torch.randn(B, D, H, W, generator=gg)
    Wx = 3.0 * torch.randn(4, R + 2 * N, D, generator=gg) / D ** 0.5
    Wdt = 4.0 * torch.randn(4, D, R, generator=gg) * R ** -0.5
    bias = torch.rand(4, D, generator=gg) * 12.0 - 8.0
    A_logs = torch.rand(4 * D, N, generator=gg) * 9.0 - 3.0
    Ds = torch.randn(4 * D, generator=gg)
    if kind == "dtpush":
        for r0 in range(0, H, 7):
            x[:, :, r0:r0 + 3] *= 40.0
        x[:, :, :, W // 2:W // 2 + 2] *= 40.0
    elif kind == "small":
        x *= 1e-3
    elif kind == "large":
        x *= 1e2
    elif kind == "zeroplane":
        x[:, D // 3] = 0.0
        if B > 1:
            x[1] = 0.0
    elif kind == "zeroD":
        Ds.zero_()
    elif kind == "zerox":
        x.zero_()
    else:
        assert kind == "trained"
    return x, Wx, Wdt, bias, A_logs, Ds


def core_eval(x, Wx, Wdt, bias, A_logs, Ds, dys, dtype):
    """SS2D.forward_core (:446-478) with the SEQUENTIAL scan definition on the CPU through autograd, in
    `dtype`: float64 = the truth, float32 = what the reference's own arithmetic gives.  -> (four outputs, six gradients)."""
    leaves = [t.detach().cpu().to(dtype).requires_grad_(True) for t in (x, Wx, Wdt, bias, A_logs, Ds)]
    x_, Wx_, Wdt_, b_, Al_, Ds_ = leaves
    B, D, H, W = x_.shape
    L, K = H * W, 4
    R, N = Wdt_.shape[2], Al_.shape[1]
    xs = torch.stack([x_.view(B, -1, L), x_.transpose(2, 3).contiguous().view(B, -1, L)], dim=1).view(B, 2, -1, L)
    xs = torch.cat([xs, torch.flip(xs, dims=[-1])], dim=1)
    x_dbl = torch.einsum("b k d l, k c d -> b k c l", xs, Wx_)
    dts, Bs, Cs = torch.split(x_dbl, [R, N, N], dim=2)
    dts = torch.einsum("b k r l, k d r -> b k d l", dts, Wdt_)
    u = xs.reshape(B, K * D, L)
    dt = F.softplus(dts.reshape(B, K * D, L) + b_.reshape(1, -1, 1))
    A = -torch.exp(Al_)
    Bf = Bs.repeat_interleave(D, dim=1)
    Cf = Cs.repeat_interleave(D, dim=1)
    h = torch.zeros(B, K * D, N, dtype=dtype)
    ys = []
    for dt_t, du_t, B_t, C_t in zip(dt.unbind(2), (dt * u).unbind(2), Bf.unbind(3), Cf.unbind(3)):
        h = torch.exp(dt_t[..., None] * A.view(1, K * D, N)) * h + du_t[..., None] * B_t
        ys.append((h * C_t).sum(-1))
    out = (torch.stack(ys, 2) + u * Ds_.view(1, -1, 1)).view(B, K, -1, L)
    inv = torch.flip(out[:, 2:4], dims=[-1]).view(B, 2, -1, L)
    wh = out[:, 1].view(B, -1, W, H).transpose(2, 3).contiguous().view(B, -1, L)
    invwh = inv[:, 1].view(B, -1, W, H).transpose(2, 3).contiguous().view(B, -1, L)
    outs = (out[:, 0], inv[:, 0], wh, invwh)
    grads = torch.autograd.grad(outs, leaves, [d.detach().cpu().to(dtype) for d in dys], allow_unused=True)
    return [o.detach() for o in outs], grads


def assert_vs_truth_ood(got, ref, truth, what):
    """Out-of-distribution bar: err(got, float64 truth) <= max(1e-4, 5 err(fp32 reference arithmetic, float64 truth)), WITHOUT
    truth_bar's 5e-4 cap: with the dt pre-activation 40x outside its trained range the reference's own fp32 gradients are up to
    3e-3 from their float64 values (tools/core_ood_report.py, profiles/r06/core_ood_report_after.txt: cancelling sums over states
    that differ by e^{+-1000}) - where fp32 itself loses the 1e-4, the bound is the reference's own loss times a small factor, never
    a fixed looser number.  The factor: the build's B / C rows of x_proj are a two-term bf16 product (2^-17; the dt_r rows carry 24
    bits since round 6) against fp32's 2^-24 - in a case that amplifies rounding 500x (reference 7e-5) the build measures 2.7e-4."""
    e_got, e_ref = max(rel_err(got, truth)), max(rel_err(ref, truth))
    assert e_got <= max(1e-4, 5.0 * e_ref), f"{what}: {e_got:.3e} vs float64 truth (fp32 reference arithmetic: {e_ref:.3e})"


OOD_SHAPES = [(1, 64, 24, 40, 16, 2), (2, 64, 33, 71, 16, 2), (1, 64, 20, 36, 32, 2), (1, 16, 19, 24, 16, 1)]


@pytest.mark.parametrize("kind", ["trained", "dtpush", "small", "large", "zeroplane", "zeroD", "zerox"])
@pytest.mark.parametrize("B,D,H,W,N,R", OOD_SHAPES)
def test_ss2d_core_forward_backward_out_of_distribution(B, D, H, W, N, R, kind):
    """The kernel the network actually runs (wm_ss2d_core_fwd / _bwd) on operands outside anything a default init produces.
    Forward: against the pinned C oracle at 1e-4; where the oracle's own fp32 arithmetic is further than 5e-5 from the float64
    evaluation of the reference formula (cancelling sums at |A| dt ~ 1e3), the bar is the truth bar: err(build, float64) <=
    max(1e-4, 5 err(fp32 reference arithmetic, float64)) (assert_vs_truth_ood).  Backward: every gradient on that bar."""
    case = ood_core_case(B, D, H, W, N, R, seed=1000 + H * 7 + W + N, kind=kind)
    L = H * W
    dys = [torch.randn(B, D, L, generator=gen(17 + i)) for i in range(4)]
    want = oracle.ss2d_core_raw(*case)
    truth_y, truth_g = core_eval(*case, dys, torch.float64)
    ref_y, ref_g = core_eval(*case, dys, torch.float32)
    args = [t.clone().requires_grad_(True) for t in cu(*case)]
    got = wm.ops.ss2d_core(*args)
    for i, (a, o, tr, rf) in enumerate(zip(got, want, truth_y, ref_y)):
        assert torch.isfinite(a).all(), f"{kind} y{i}: non-finite values"
        if kind == "zerox":
            assert float(a.abs().max()) == 0.0 and float(o.abs().max()) == 0.0
            continue
        e_o = max(rel_err(o, tr))
        if e_o <= 5e-5:
            assert_close(a, o, TOL, f"{kind} core y{i} vs oracle")
        assert_vs_truth_ood(a, rf if max(rel_err(rf, tr)) > e_o else o, tr, f"{kind} core y{i}")
    merged = wm.ops.ss2d_core(*cu(*case), merged=True)
    if kind != "zerox":
        assert_vs_truth_ood(merged, sum(ref_y), sum(truth_y), f"{kind} merged")
    grads = torch.autograd.grad(got, args, cu(*dys))
    for a, rf, tr, nm in zip(grads, ref_g, truth_g, ("dx", "dWx", "dWdt", "dbias", "dA_logs", "dDs")):
        assert torch.isfinite(a).all(), f"{kind} {nm}: non-finite values"
        if float(tr.abs().max()) == 0.0:
            assert float(a.abs().max()) <= 1e-6, f"{kind} {nm}: expected zeros"
            continue
        assert_vs_truth_ood(a, rf, tr, f"{kind} {nm} {(B, D, H, W, N, R)}")


def perturb_like_trained(module, seed):
    """Every parameter of `module` off its init: SS2D's A_logs ~ U[-3, 6], dt_projs_bias ~ U[-8, 4], Ds ~ N(0, 1), x_proj x3,
    dt_projs x4; every other tensor multiplicative and additive noise of a quarter of its own scale."""
    g = gen(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            leaf = name.rsplit(".", 1)[-1]
            if leaf == "A_logs":
                p.copy_(torch.rand(p.shape, generator=g) * 9.0 - 3.0)
            elif leaf == "dt_projs_bias":
                p.copy_(torch.rand(p.shape, generator=g) * 12.0 - 8.0)
            elif leaf == "Ds":
                p.copy_(torch.randn(p.shape, generator=g))
            elif leaf == "x_proj_weight":
                p.mul_(3.0)
            elif leaf == "dt_projs_weight":
                p.mul_(4.0)
            else:
                scale = float(p.abs().mean()) + 1e-3
                p.mul_(1.0 + 0.25 * torch.randn(p.shape, generator=g)).add_(0.25 * scale * torch.randn(p.shape, generator=g))


@pytest.mark.parametrize("H,W,d_state", [(24, 40, 16), (33, 32, 16), (16, 24, 32)])
def test_lfss_block_out_of_distribution_vs_cpu_oracle(H, W, d_state):
    """One LFSSBlock with trained-like parameters (perturb_like_trained), inference kernels and the training path (fused core
    forward + backward), against the SAME block on the host with the CPU oracle as hot-path backend: output at 1e-4, input and
    parameter gradients at the truth-free bar 2e-4 (two fp32 implementations of sums over L positions)."""
    import copy
    torch.manual_seed(3)
    blk_cpu = arch.LFSSBlock(32, d_state=d_state, expand=2.0)
    perturb_like_trained(blk_cpu, seed=H + W)
    x_cpu = (torch.randn(2, H * W, 32, generator=gen(9)) * 2.0)
    x_cpu[:, H * W // 3:H * W // 3 + W] *= 30.0                      # one image row far outside the rest
    blk = copy.deepcopy(blk_cpu).to(DEV)
    with oracle_backend.ops_backend(oracle), torch.no_grad():
        want = blk_cpu.eval()(x_cpu, [H, W])
    with torch.no_grad():
        got = blk.eval()(x_cpu.to(DEV), [H, W])
    assert_close(got, want, TOL, "OOD LFSSBlock, inference kernels")
    gy = torch.randn(want.shape, generator=gen(10))
    xc = x_cpu.clone().requires_grad_(True)
    with oracle_backend.ops_backend(oracle):
        out_c = blk_cpu.train()(xc, [H, W])
        ref = torch.autograd.grad(out_c, [xc] + list(blk_cpu.parameters()), gy)
    xg = x_cpu.to(DEV).requires_grad_(True)
    out_g = blk.train()(xg, [H, W])
    assert_close(out_g, out_c, TOL, "OOD LFSSBlock, training path output")
    grads = torch.autograd.grad(out_g, [xg] + list(blk.parameters()), gy.to(DEV))
    names = ["dx"] + [n for n, _ in blk.named_parameters()]
    for a, b, nm in zip(grads, ref, names):
        assert_close(a, b, 2e-4, f"OOD LFSSBlock grad {nm}")


def test_network_256_out_of_distribution_vs_cpu_oracle_network():
    """BASELINE config 1's 256 x 256 input through the shipped configuration with EVERY parameter off its init (trained-like SS2D
    parameters in all 14 LFSSBlocks) against the CPU-oracle network: rel-l2 and max-abs <= 1e-4."""
    import copy
    torch.manual_seed(0)
    net_cpu = wm.WaveMamba(in_chn=3, wf=32, n_l_blocks=[1, 2, 4], n_h_blocks=[1, 1, 2], ffn_scale=2.0).eval()
    perturb_like_trained(net_cpu, seed=5)
    img = torch.rand(1, 3, 256, 256, generator=gen(1234))
    cores = oracle.usable_cpus(cap=1 << 20)
    torch.set_num_threads(cores); oracle.set_num_threads(cores)
    with oracle_backend.ops_backend(oracle), torch.no_grad():
        want = net_cpu(img)
    net = copy.deepcopy(net_cpu).to(DEV)
    with torch.no_grad():
        got = net(img.to(DEV))
    assert torch.isfinite(got).all()
    l2, mx = rel_err(got.cpu(), want)
    print(f"OOD 256 x 256 network vs CPU-oracle network: rel-l2 {l2:.3e}, max-abs / max-abs {mx:.3e}")
    assert_close(got, want, TOL, "OOD 256 x 256 network")


def test_ss2d_core_uhd_level2_against_unfused():
    """UHD level-2 map (544 x 960, L = 522,240)."""
    core_vs_unfused_and_oracle_subset(544, 960, 16, seed=3, channels=[0, 31, 63], what="UHD-L2")


def test_ss2d_core_uhd_level1_full_size():
    """BASELINE config 2, the launch that dominates the bench: UHD level-1 map (1088 x 1920, L = 2,088,960) through the
    product path's fused core."""
    core_vs_unfused_and_oracle_subset(1088, 1920, 16, seed=4, channels=[0, 21, 42, 63], what="UHD-L1")


def test_ss2d_core_uhd_level1_full_size_every_channel_of_a_column_direction():
    """The same launch with ALL 64 channels of direction 3 (column-major, reversed: the direction with the most index arithmetic)
    against the CPU oracle's scan, 4 channels of the other three."""
    core_vs_unfused_and_oracle_subset(1088, 1920, 16, seed=6, channels=[5, 26, 47, 60], what="UHD-L1 all channels of k=3",
                                      full_direction=3)


def test_ss2d_core_config5_full_size():
    """BASELINE config 5: d_state 32 on a 2048 x 2048 tile (L = 4,194,304, 4x the UHD level-1 sequence length)."""
    core_vs_unfused_and_oracle_subset(2048, 2048, 32, seed=5, channels=[7, 56], what="config 5")


def test_lfss_block_config5_full_size():
    """BASELINE config 5 as worded: LFSSBlock(32, d_state=32) on (1, 4194304, 32) with x_size [2048, 2048] - the fused HIP block
    against the SAME block on the host with the CPU oracle as hot-path backend (the pinned C restatement of SS2D.forward_core at
    d_state 32 over the whole 2048 x 2048 map, PyTorch-CPU for the rest; about a minute on the box's host cores) - every
    element of the output, not a self-comparison with another HIP path - plus run-to-run bit identity."""
    torch.manual_seed(0)
    blk_cpu = arch.LFSSBlock(32, d_state=32, expand=2.0).eval()
    x_cpu = torch.randn(1, 2048 * 2048, 32, generator=gen(12))
    cores = oracle.usable_cpus(cap=1 << 20)
    torch.set_num_threads(cores); oracle.set_num_threads(cores)
    with oracle_backend.ops_backend(oracle), torch.no_grad():
        want = blk_cpu(x_cpu, [2048, 2048])
    import copy
    blk = copy.deepcopy(blk_cpu).to(DEV)
    x = x_cpu.to(DEV)
    with torch.no_grad():
        assert blk._fused_ok(x, 2048)
        fused = blk(x, [2048, 2048])
        again = blk(x, [2048, 2048])
    assert torch.equal(fused, again)
    l2, mx = rel_err(fused.cpu(), want)
    print(f"config 5 LFSSBlock vs the CPU-oracle block: rel-l2 {l2:.3e}, max-abs / max-abs {mx:.3e}")
    assert_close(fused.cpu(), want, TOL, "config 5 LFSSBlock vs CPU oracle")


@pytest.fixture(scope="module")
def uhd_cpu_oracle():
    """The padded UHD frame of inference_wavemamba.py:28-36, :99-113 and its forward through the same network on the host
    with the CPU oracle as hot-path backend (about a minute of CPU time; shared by the fp32 and the bf16-storage test)."""
    import bench
    img = torch.rand(1, 3, 2160, 3840, generator=gen(1234))
    x = bench.pad_to(img)
    assert x.shape[-2:] == (2176, 3840)
    cores = oracle.usable_cpus(cap=1 << 20)
    torch.set_num_threads(cores); oracle.set_num_threads(cores)
    net = bench.build_model("cpu")
    with oracle_backend.ops_backend(oracle), torch.no_grad():
        want = net.restoration_network(x)
    return x, want, net


def test_uhd_forward_full_size_vs_cpu_oracle_network(uhd_cpu_oracle):
    """BASELINE config 2 end to end on the product path (fp32): ONE full 2176 x 3840 forward against the CPU-oracle
    network: rel-l2 <= 1e-4 and |dPSNR| <= 1e-3 dB after the reference's uint8 quantisation."""
    import bench
    x, want, net = uhd_cpu_oracle
    net = bench.build_model(DEV)
    with torch.no_grad():
        got = net.restoration_network(x.to(DEV)).cpu()
    l2, mx = rel_err(got, want)
    print(f"UHD forward vs CPU-oracle network: rel-l2 {l2:.3e}, max-abs / max-abs {mx:.3e}")
    assert_close(got, want, TOL, "UHD forward")
    tgt = torch.rand(1, 3, 2160, 3840, generator=gen(4321))
    crop = lambda t: t[:, :, :2160, :3840]
    assert abs(bench.psnr_u8(crop(got), tgt) - bench.psnr_u8(crop(want), tgt)) <= 1e-3


def test_uhd_forward_bf16_storage_vs_cpu_oracle_network(uhd_cpu_oracle):
    """BASELINE config 2 AS WORDED (bf16) at its real size: the bf16-storage mode (bf16 planes between the kernels of
    every LFSSBlock, fp32 arithmetic inside) on the full 2176 x 3840 frame against the fp32 CPU-oracle network.  No 1e-4
    bar can hold at 8 mantissa bits (the reference itself under torch.autocast(bfloat16) is 59.6 dB from its fp32
    output); stated bars: PSNR >= 60 dB on the [0, 1] outputs, >= 55 dB after uint8 quantisation
    (measured on MI355X: 70.1 / 60.4 dB)."""
    import bench
    from wave_mamba_amd import inference
    x, want, _ = uhd_cpu_oracle
    net = bench.build_model(DEV)
    prev = wm.ops.set_plane_dtype(torch.bfloat16)
    try:
        with torch.no_grad():
            got = net.restoration_network(x.to(DEV)).cpu()
    finally:
        wm.ops.set_plane_dtype(prev)
    crop = lambda t: t[:, :, :2160, :3840]
    mse = float((crop(got).clamp(0, 1) - crop(want).clamp(0, 1)).double().pow(2).mean())
    psnr = float(10 * torch.log10(torch.tensor(1.0 / mse)))
    psnr8 = inference.psnr_uint8(inference.to_uint8(crop(got)), inference.to_uint8(crop(want)))
    print(f"bf16 storage at UHD vs fp32 CPU-oracle network: PSNR {psnr:.1f} dB, after uint8 {psnr8:.1f} dB, "
          f"rel-l2 {rel_err(got, want)[0]:.2e}")
    assert psnr >= 60.0 and psnr8 >= 55.0


def test_core_abi_error_codes_from_real_calls():
    """The C ABI's status codes on a GPU box, through ctypes: short workspace -> WM_EWORKSPACE, misaligned pointer ->
    WM_EALIGN, missing pointer -> WM_ENULL, out-of-range shape -> WM_EUNSUPPORTED / WM_EINVAL; a good call -> WM_OK and
    the message table answers for every code."""
    from wave_mamba_amd import _lib
    lib = _lib.load()
    B, D, H, W, N, R = 1, 64, 32, 48, 16, 2
    x, Wx, Wdt, bias, A_logs, Ds = cu(*random_core_case(B, D, H, W, N, R, seed=1))
    ys = torch.empty(4, B, D, H * W, device=DEV)
    need = lib.wm_ss2d_core_fwd_workspace_bytes(B, D, H, W, N, R, 0)
    ws = torch.empty(need + 64, dtype=torch.uint8, device=DEV)
    p = lambda t: t.data_ptr()
    yp = [p(ys[i]) for i in range(4)]

    def call(xp=p(x), wsp=p(ws), wsb=need, y0=yp[0], n=N, h=H, prep=None, pd=0):
        return lib.wm_ss2d_core_fwd(xp, p(Wx), p(Wdt), p(bias), p(A_logs), p(Ds), y0, yp[1], yp[2], yp[3], 0, wsp, wsb,
                                    prep, B, D, h, W, n, R, pd, torch.cuda.current_stream().cuda_stream)
    assert call() == 0
    # the prepared-parameters form gives the same bits as the self-preparing call
    y_self = ys.clone()
    pb = torch.empty(lib.wm_ss2d_core_prep_bytes(N), dtype=torch.uint8, device=DEV)
    assert lib.wm_ss2d_core_prep(p(Wx), p(Wdt), p(bias), p(A_logs), p(Ds), p(pb), D, N, R,
                                 torch.cuda.current_stream().cuda_stream) == 0
    ys.zero_()
    assert call(prep=p(pb)) == 0 and torch.equal(ys, y_self)
    assert call(prep=p(pb) + 4) == -3                     # WM_EALIGN (prepared buffer)
    assert lib.wm_ss2d_core_prep_bytes(64) == 0
    assert call(wsb=need - 1) == -4                       # WM_EWORKSPACE
    assert call(wsp=p(ws) + 4) == -3                      # WM_EALIGN (workspace)
    assert call(xp=p(x) + 2) == -3                        # WM_EALIGN (x not even element-aligned; + 4 would run: element-wise tiles)
    assert call(xp=p(x) + 4, pd=1) == -3                  # WM_EALIGN (bf16 planes exist with 16-byte tile accesses only)
    assert call(y0=None) == -2                            # WM_ENULL
    assert call(n=64) == -5                               # WM_EUNSUPPORTED
    assert call(h=-1) == -1                               # WM_EINVAL
    assert lib.wm_ss2d_core_fwd_workspace_bytes(B, D, H, W, 64, R, 0) == 0
    for code in (0, -1, -2, -3, -4, -5, -6):
        assert lib.wm_strerror(code)
    torch.cuda.synchronize()
    # the drop-in scan: short workspace / misaligned workspace
    u = torch.randn(1, 64, 4096, device=DEV); A = -torch.rand(64, 16, device=DEV); Bc = torch.randn(1, 1, 16, 4096, device=DEV)
    out = torch.empty_like(u)
    nb = lib.wm_selscan_fwd_workspace_bytes(1, 64, 4096, 16, 1)
    w2 = torch.empty(nb + 64, dtype=torch.uint8, device=DEV)
    args = lambda wsp, wsb: (p(u), p(u), p(A), p(Bc), p(Bc), None, None, None, p(out), None, wsp, wsb, 1, 64, 4096, 16, 1, 1,
                             torch.cuda.current_stream().cuda_stream)
    assert lib.wm_selscan_fwd(*args(p(w2), nb)) == 0
    if nb:
        assert lib.wm_selscan_fwd(*args(p(w2), nb - 1)) == -4
        assert lib.wm_selscan_fwd(*args(p(w2) + 4, nb)) == -3
    torch.cuda.synchronize()


# ------------------------------------------------------------------------------------------------
# whole LFSSBlock on the HIP path (reference LFSSBlock.forward, :520-528)
# ------------------------------------------------------------------------------------------------
def test_lfss_block_golden(golden):
    g = golden("lfss_block")            # input/output/weights captured from the reference's LFSSBlock
    blk = arch.LFSSBlock(32, expand=2.0).eval()
    blk.load_state_dict({k[2:]: v for k, v in g.items() if k.startswith("p.")}, strict=True)
    blk = blk.to(DEV)
    with torch.no_grad():
        assert blk._fused_ok(g["x"].to(DEV))
        y = blk(g["x"].to(DEV), [8, 12])
    assert_close(y, g["y"], TOL, "LFSSBlock (fused HIP path)")


@pytest.mark.parametrize("B,H,W,nchw", [(1, 40, 96, True), (2, 33, 32, False), (1, 272, 480, True), (2, 5, 64, False), (1, 1, 32, True)])
def test_lfss_block_recomputed_gate_bit_identical(B, H, W, nchw):
    """The block with the gate z recomputed inside lfss_mid from the tokens (wm_lfss_mid_rz_fwd; lfss_in writes the x half only)
    against the block with z written by lfss_in and read back: the same matrix instructions in the same order - equal bit for bit
    on fp32 planes; bf16 planes: the recomputed gate skips one bf16 rounding (compared at the bf16-storage bar)."""
    torch.manual_seed(H)
    blk = arch.LFSSBlock(32, expand=2.0).eval().to(DEV)
    with torch.no_grad():
        for p in blk.parameters():
            p.add_(0.05 * torch.randn_like(p))
        x = torch.randn(B, 32, H, W, device=DEV) if nchw else torch.randn(B, H * W, 32, device=DEV)
        outs = {}
        for rz in (True, False):
            prev, wm.ops._RECOMPUTE_Z = wm.ops._RECOMPUTE_Z, rz
            try:
                outs[rz] = wm.ops.lfss_block_forward(x, (H, W), blk, tok_nchw=nchw, out_nchw=nchw)
                if W % 4 == 0:
                    pd = wm.ops.set_plane_dtype(torch.bfloat16)
                    try:
                        outs[(rz, "bf16")] = wm.ops.lfss_block_forward(x, (H, W), blk, tok_nchw=nchw, out_nchw=nchw)
                    finally:
                        wm.ops.set_plane_dtype(pd)
            finally:
                wm.ops._RECOMPUTE_Z = prev
    assert torch.equal(outs[True], outs[False]), f"recomputed gate: max abs difference {float((outs[True] - outs[False]).abs().max()):.3e}"
    if (True, "bf16") in outs:
        assert_close(outs[(True, "bf16")], outs[False], 2e-2, "bf16 planes, recomputed gate vs fp32")
        assert_close(outs[(True, "bf16")], outs[(False, "bf16")], 1e-2, "bf16 planes, recomputed vs stored gate")


@pytest.mark.parametrize("C,H,W", [(32, 40, 72), (32, 33, 31), (32, 5, 7), (16, 17, 23), (8, 64, 64)])
def test_lfss_block_fused_vs_module_path(C, H, W):
    """fused block kernels vs the same block evaluated through the PyTorch modules + HIP scan."""
    torch.manual_seed(C)
    blk = arch.LFSSBlock(C, expand=2.0).eval().to(DEV)
    with torch.no_grad():
        for p in blk.parameters():
            p.add_(0.05 * torch.randn_like(p))
        x = torch.randn(2, H * W, C, device=DEV)
        fused = blk(x, [H, W])
        saved = arch.LFSSBlock._fused_ok
        arch.LFSSBlock._fused_ok = lambda self, t, width=None, height=None: False
        try:
            ref = blk(x, [H, W])
        finally:
            arch.LFSSBlock._fused_ok = saved
    assert_close(fused, ref, TOL, f"LFSSBlock C={C}")


@pytest.mark.parametrize("B,H,W", [(1, 40, 96), (2, 3, 32), (1, 1, 64), (2, 2, 32), (1, 17, 480), (1, 5, 160),
                                   (1, 9, 128), (2, 7, 192), (1, 2, 64), (1, 34, 960),       # W % 64 == 0: the banded one-row form
                                   (1, 1025, 1024), (2, 513, 1088)])                          # >= 2^20 positions: the row-window form (odd H: a short last band)
@pytest.mark.parametrize("nchw", [False, True])
def test_lfss_out_with_depthwise_conv_folded_in(B, H, W, nchw):
    """wm_lfss_out_conv_fwd (the ffn's depth-wise 3x3 inside the closing kernel) against
    wm_dwconv3x3_fwd + wm_lfss_out_fwd: BIT-identical on fp32 planes - interior tiles, first / last rows, one-row and
    two-row maps, a row that is one tile (both column edges in it), batch 2, an odd number of 32-position tiles - and
    both against the fp64 composition of reference :226-230 (conv2 -> gelu(x1) * x2 -> conv3, then * skip + tok1)."""
    from wave_mamba_amd import _lib
    from wave_mamba_amd.ops import _ptr, _stream, check
    lib = _lib.load()
    C, D, L = 32, 64, H * W
    g = torch.Generator(device=DEV); g.manual_seed(B * 1000 + H * 10 + W)
    rn = lambda *s: torch.randn(*s, device=DEV, generator=g)
    f, tok1 = rn(B, D, H, W), rn(B, L, C)
    w2, b2, w3, b3, sk = rn(D, 1, 3, 3) * 0.3, rn(D) * 0.2, rn(C, C, 1, 1) * 0.2, rn(C) * 0.2, rn(C) * 0.3 + 1
    shape = (B, C, H, W) if nchw else (B, L, C)
    fused, unfused = torch.empty(shape, device=DEV), torch.empty(shape, device=DEV)
    check(lib.wm_lfss_out_conv_fwd(_ptr(f), _ptr(w2), _ptr(b2), _ptr(tok1), _ptr(w3), _ptr(b3), _ptr(sk), _ptr(fused),
                                   int(nchw), B, H, W, C, 0, _stream()), "wm_lfss_out_conv_fwd")
    fc = wm.ops.dwconv3x3(f, w2, b2, "none")
    check(lib.wm_lfss_out_fwd(_ptr(fc), _ptr(tok1), _ptr(w3), _ptr(b3), _ptr(sk), _ptr(unfused), int(nchw), B, L, C, 0,
                              _stream()), "wm_lfss_out_fwd")
    assert torch.equal(fused, unfused)
    fc64 = F.conv2d(f.double(), w2.double(), b2.double(), padding=1, groups=D)
    gv = F.gelu(fc64[:, :C]) * fc64[:, C:]
    want = F.conv2d(gv, w3.double(), b3.double()) + (tok1.double() * sk.double()).transpose(1, 2).reshape(B, C, H, W)
    if not nchw:
        want = want.reshape(B, C, L).transpose(1, 2)
    assert_close(fused, want.float(), 1e-5, "lfss_out with conv2 folded in vs fp64")
    # no bias, bf16 planes (fp32 arithmetic on the rounded input)
    fb16 = f.bfloat16()
    check(lib.wm_lfss_out_conv_fwd(_ptr(fb16), _ptr(w2), None, _ptr(tok1), _ptr(w3), _ptr(b3), _ptr(sk), _ptr(fused),
                                   int(nchw), B, H, W, C, 1, _stream()), "wm_lfss_out_conv_fwd bf16")
    fc64 = F.conv2d(f.bfloat16().double(), w2.double(), None, padding=1, groups=D)
    gv = F.gelu(fc64[:, :C]) * fc64[:, C:]
    want = F.conv2d(gv, w3.double(), b3.double()) + (tok1.double() * sk.double()).transpose(1, 2).reshape(B, C, H, W)
    if not nchw:
        want = want.reshape(B, C, L).transpose(1, 2)
    assert_close(fused, want.float(), 1e-5, "bf16 planes, no conv2 bias")
    assert lib.wm_lfss_out_conv_fwd(_ptr(f), _ptr(w2), _ptr(b2), _ptr(tok1), _ptr(w3), _ptr(b3), _ptr(sk), _ptr(fused),
                                    int(nchw), B, H, W - 4, C, 0, _stream()) == -5       # W % 32 != 0: WM_EUNSUPPORTED


@pytest.mark.parametrize("B,L", [(1, 64), (2, 1000), (3, 37), (1, 4097)])
@pytest.mark.parametrize("nchw", [False, True])
def test_lfss_glue_kernels_c32_vs_fp64(B, L, nchw):
    """The three glue kernels at the shipped width (C = 32: projections on the fp32 matrix cores), called through the
    C ABI with both token layouts and ragged group counts, against the fp64 composition of reference :483-494,
    :524-526, :226-230."""
    from wave_mamba_amd import _lib
    from wave_mamba_amd.ops import _ptr, _stream, check
    lib = _lib.load()
    C, D = 32, 64
    g = torch.Generator(device=DEV); g.manual_seed(B * 131 + L)
    rn = lambda *s: torch.randn(*s, device=DEV, generator=g)
    ln1w, ln1b, ln2w, ln2b = rn(C) * 0.2 + 1, rn(C) * 0.2, rn(C) * 0.2 + 1, rn(C) * 0.2
    onw, onb = rn(D) * 0.2 + 1, rn(D) * 0.2
    Win, Wout, W1, b1, W3, b3 = rn(2 * D, C) / 6, rn(C, D) / 8, rn(D, C) / 6, rn(D) * 0.2, rn(C, C) / 6, rn(C) * 0.2
    sk1, sk2 = rn(C) * 0.2 + 1, rn(C) * 0.2 + 1
    tokens = rn(B, L, C)                                                    # logical (B, L, C)
    tok = tokens.transpose(1, 2).contiguous() if nchw else tokens           # as laid out for the kernels
    ysum, zz, fc = rn(B, D, L) * 2, rn(B, D, L) * 2, rn(B, D, L)
    F = torch.nn.functional
    d = lambda t: t.double()
    # lfss_in
    x = torch.empty(B, D, L, device=DEV); z = torch.empty(B, D, L, device=DEV)
    check(lib.wm_lfss_in_fwd(_ptr(tok), int(nchw), _ptr(ln1w), _ptr(ln1b), 1e-5, _ptr(Win), _ptr(x), _ptr(z), B, L, C,
                             0, _stream()), "in")
    xz = F.linear(F.layer_norm(d(tokens), (C,), d(ln1w), d(ln1b), 1e-5), d(Win)).transpose(1, 2)
    assert_close(x, xz[:, :D], TOL, "lfss_in x"); assert_close(z, xz[:, D:], TOL, "lfss_in z")
    # lfss_mid
    tok1 = torch.empty(B, L, C, device=DEV); f = torch.empty(B, D, L, device=DEV)
    check(lib.wm_lfss_mid_fwd(_ptr(ysum), 1, 0, _ptr(zz), _ptr(tok), int(nchw), _ptr(onw), _ptr(onb), 1e-5, _ptr(Wout), _ptr(sk1),
                              _ptr(ln2w), _ptr(ln2b), 1e-5, _ptr(W1), _ptr(b1), _ptr(tok1), _ptr(f), B, L, C, 0, _stream()), "mid")
    yy = F.layer_norm(d(ysum).transpose(1, 2), (D,), d(onw), d(onb), 1e-5) * F.silu(d(zz).transpose(1, 2))
    t1 = d(tokens) * d(sk1) + F.linear(yy, d(Wout))
    ff = F.linear(F.layer_norm(t1, (C,), d(ln2w), d(ln2b), 1e-5), d(W1), d(b1)).transpose(1, 2)
    assert_close(tok1, t1, TOL, "lfss_mid tok1"); assert_close(f, ff, TOL, "lfss_mid f")
    # lfss_out
    out = torch.empty((B, C, L) if nchw else (B, L, C), device=DEV)
    check(lib.wm_lfss_out_fwd(_ptr(fc), _ptr(tok1), _ptr(W3), _ptr(b3), _ptr(sk2), _ptr(out), int(nchw), B, L, C, 0, _stream()),
          "out")
    gg = (F.gelu(d(fc[:, :C])) * d(fc[:, C:])).transpose(1, 2)
    o = d(tok1) * d(sk2) + F.linear(gg, d(W3), d(b3))
    assert_close(out.transpose(1, 2) if nchw else out, o, TOL, "lfss_out")


def test_lfss_block_d_state_32_on_the_fused_core():
    """BASELINE config 5 flavour: LFSSBlock(32, d_state=32) takes the fused HIP block path (N = 32 instantiation of
    the SS2D core).  Checked against the same block on the CPU oracle backend; a width that is not a multiple of 4
    takes the same fused path (element-wise tile accesses), same check."""
    torch.manual_seed(5)
    blk = arch.LFSSBlock(32, d_state=32, expand=2.0).eval()
    for (H, W) in ((48, 40), (20, 30)):
        x = torch.randn(1, H * W, 32, generator=gen(9))
        prev = oracle_backend.set_ops_backend(oracle)
        try:
            with torch.no_grad():
                want = blk.cpu()(x, [H, W])
        finally:
            oracle_backend.set_ops_backend(prev)
        blk = blk.to(DEV)
        with torch.no_grad():
            assert blk._fused_ok(x.to(DEV), W, H)
            got = blk(x.to(DEV), [H, W])
        assert_close(got, want, TOL, f"LFSSBlock d_state=32 {H}x{W}")


# ------------------------------------------------------------------------------------------------
# bf16-storage mode (BASELINE config 2 as worded): bf16 planes between the LFSSBlock kernels, fp32 inside
# ------------------------------------------------------------------------------------------------
BF16_ULP = 2.0 ** -8          # one bf16 rounding, relative


@pytest.mark.parametrize("H,W,N", [(32, 48, 16), (40, 136, 16), (24, 40, 32)])
def test_ss2d_core_bf16_planes(H, W, N):
    """bf16 x in / bf16 y out = the fp32 core on the same (bf16-representable) x, rounded once to bf16."""
    case = cu(*random_core_case(1, 64, H, W, N, 2, seed=H + W))
    xb = case[0].bfloat16()
    want = wm.ops.ss2d_core(xb.float(), *case[1:])
    got = wm.ops.ss2d_core(xb, *case[1:])
    for i, (a, b) in enumerate(zip(got, want)):
        assert a.dtype == torch.bfloat16
        assert torch.equal(a, b.bfloat16()), f"y{i}: bf16 core differs from the rounded fp32 core"
    gm, wmg = wm.ops.ss2d_core(xb, *case[1:], merged=True), sum(t.bfloat16().float() for t in want)
    assert_close(gm.float(), wmg, 2 * BF16_ULP, "merged bf16")


@pytest.mark.parametrize("act", ["none", "silu", "gelu"])
def test_dwconv3x3_bf16_planes(act):
    x = torch.randn(2, 64, 20, 36, generator=gen(4)).to(DEV).bfloat16()
    w = torch.randn(64, 1, 3, 3, generator=gen(5)).to(DEV) / 3
    b = torch.randn(64, generator=gen(6)).to(DEV)
    got = wm.ops.dwconv3x3(x, w, b, act)
    want = wm.ops.dwconv3x3(x.float(), w, b, act)
    assert got.dtype == torch.bfloat16 and torch.equal(got, want.bfloat16())


def test_bf16_storage_mode_block_and_network(golden):
    """The mode end to end: an LFSSBlock and the shipped network at 256 x 256 with bf16 planes against the fp32 path.
    No 1e-4 bar can hold at 8 mantissa bits; the stated bars are rel-l2 <= 2e-2 for one block on unit-scale random
    tokens and PSNR >= 40 dB for the network (measured values are printed; the reference under torch.autocast(bf16)
    sits at 59.6 dB on its own CPU path)."""
    from wave_mamba_amd import inference
    torch.manual_seed(3)
    blk = arch.LFSSBlock(32, expand=2.0).eval().to(DEV)
    x = torch.randn(1, 64 * 96, 32, generator=gen(7)).to(DEV)
    with torch.no_grad():
        ref = blk(x, [64, 96])
        prev = wm.ops.set_plane_dtype(torch.bfloat16)
        try:
            got = blk(x, [64, 96])
        finally:
            wm.ops.set_plane_dtype(prev)
    rel = float((got - ref).norm() / ref.norm())
    assert got.dtype == torch.float32 and 0 < rel <= 2e-2, f"block rel-l2 {rel:.3e}"
    torch.manual_seed(0)
    net = wm.WaveMamba(in_chn=3, wf=32, n_l_blocks=[1, 2, 4], n_h_blocks=[1, 1, 2], ffn_scale=2.0).eval().to(DEV)
    img = torch.rand(1, 3, 256, 256, generator=gen(1234)).to(DEV)
    r = inference.bench_bf16_storage(net, img, steps=1, warmup=1)
    print(f"bf16 storage: block rel-l2 {rel:.2e}; network PSNR {r['psnr_vs_fp32_db']:.1f} dB (uint8 {r['psnr_u8_vs_fp32_db']:.1f} dB)")
    assert r["psnr_vs_fp32_db"] >= 40.0
    assert wm.ops.get_plane_dtype() == torch.float32


# ------------------------------------------------------------------------------------------------
# HFE-branch helpers ("next" row): floating-point kernels of standard ops -> torch fp32
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("C,H,W", [(32, 33, 47), (16, 8, 8), (8, 5, 130)])
def test_layernorm2d_vs_torch(C, H, W):
    ln = arch.LayerNorm2d(C)
    with torch.no_grad():
        ln.weight.copy_(torch.randn(C, generator=gen(1)))
        ln.bias.copy_(torch.randn(C, generator=gen(2)))
    x = torch.randn(2, C, H, W, generator=gen(3)) * 3 + 1
    mu = x.mean(1, keepdim=True)
    var = (x - mu).pow(2).mean(1, keepdim=True)
    want = ln.weight.view(1, -1, 1, 1) * ((x - mu) / (var + ln.eps).sqrt()) + ln.bias.view(1, -1, 1, 1)
    with torch.no_grad():
        got = ln.to(DEV)(x.to(DEV))
    assert_close(got, want.detach(), 1e-5, "LayerNorm2d")


@pytest.mark.parametrize("shape", [(2, 64, 40, 72), (1, 5, 7, 9), (2, 96, 33, 260),
                                   # narrow maps: 4 / 2 planes side by side in a wave (a plane count that does not fill the last group,
                                   # widths below the lane group, one strip and several)
                                   (8, 64, 64, 64), (3, 7, 37, 64), (1, 5, 16, 48), (2, 3, 70, 128), (1, 9, 20, 100), (2, 32, 128, 128)])
def test_dwconv3x3_gradients_vs_torch_autograd(shape):
    import torch.nn.functional as F
    B, C, H, W = shape
    gg = gen(H * W)
    x = torch.randn(*shape, generator=gg)
    w = torch.randn(C, 1, 3, 3, generator=gg) * 0.3
    b = torch.randn(C, generator=gg)
    gy = torch.randn(*shape, generator=gg)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    want = torch.autograd.grad(F.conv2d(xr, wr, br, padding=1, groups=C), (xr, wr, br), gy)
    xg, wg, bg = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    got = torch.autograd.grad(wm.ops.dwconv3x3_train(xg, wg, bg), (xg, wg, bg), gy.to(DEV))
    for name, a, r in zip(("dx", "dW", "db"), got, want):
        assert_close(a, r, 2e-5, f"dwconv {name}")


@pytest.mark.parametrize("C,H,W", [(32, 33, 47), (16, 8, 8), (8, 5, 130), (64, 33, 47), (64, 64, 64), (32, 128, 128), (64, 1, 1), (32, 3, 5)])
def test_layernorm2d_gradients_vs_torch_autograd(C, H, W):
    x = torch.randn(2, C, H, W, generator=gen(1)) * 2 + 0.5
    w = torch.randn(C, generator=gen(2))
    b = torch.randn(C, generator=gen(3))
    gy = torch.randn(2, C, H, W, generator=gen(4))

    def ref(x_, w_, b_):
        mu = x_.mean(1, keepdim=True)
        var = (x_ - mu).pow(2).mean(1, keepdim=True)
        return w_.view(1, -1, 1, 1) * ((x_ - mu) / (var + 1e-6).sqrt()) + b_.view(1, -1, 1, 1)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    want = torch.autograd.grad(ref(xr, wr, br), (xr, wr, br), gy)
    xg, wg, bg = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    got = torch.autograd.grad(wm.ops.layernorm2d_train(xg, wg, bg, 1e-6), (xg, wg, bg), gy.to(DEV))
    for name, a, r in zip(("gx", "dw", "db"), got, want):
        assert_close(a, r, 5e-5, f"LayerNorm2d {name}")


@pytest.mark.parametrize("B,C,L", [(1, 32, 70000), (2, 32, 4099), (1, 16, 257), (3, 8, 64), (1, 32, 5)])
def test_gram_vs_torch(B, C, L):
    x = torch.randn(B, C, L, generator=gen(L)) + 0.3
    y = torch.randn(B, C, L, generator=gen(L + 1)) - 0.2
    G, nx, ny = wm.ops.gram(x.to(DEV), y.to(DEV))
    assert_close(G, (x.double() @ y.double().transpose(1, 2)).float(), 2e-5, "gram")
    assert_close(nx, x.double().pow(2).sum(-1).float(), 2e-5, "|x|^2")
    assert_close(ny, y.double().pow(2).sum(-1).float(), 2e-5, "|y|^2")


@pytest.mark.parametrize("B,C,L", [(2, 32, 4096), (8, 32, 65536), (3, 16, 1000)])
def test_gram_train_gradients_vs_normalize_path_fp64(B, C, L):
    """The training form of the transposed attention's logits - gram_train + division by the norms on the (B, C, C) result -
    against the reference's F.normalize(q) @ F.normalize(k)^T (:783-786) under autograd in float64: value and both gradients."""
    q = (torch.randn(B, C, L, generator=gen(L)) + 0.1).to(DEV).requires_grad_(True)
    k = (torch.randn(B, C, L, generator=gen(L + 7)) - 0.2).to(DEV).requires_grad_(True)
    wgt = torch.randn(B, C, C, generator=gen(5)).to(DEV)
    G, nq, nk = wm.ops.gram_train(q, k)
    attn = G / (nq.sqrt().clamp_min(1e-12).unsqueeze(2) * nk.sqrt().clamp_min(1e-12).unsqueeze(1))
    (attn * wgt).sum().backward()
    q64, k64 = q.detach().double().requires_grad_(True), k.detach().double().requires_grad_(True)
    ref = torch.nn.functional.normalize(q64, dim=-1) @ torch.nn.functional.normalize(k64, dim=-1).transpose(1, 2)
    (ref * wgt.double()).sum().backward()
    assert_close(attn.detach(), ref.detach().float(), 2e-5, "normalized gram")
    assert_close(q.grad, q64.grad.float(), 5e-5, "d q")
    assert_close(k.grad, k64.grad.float(), 5e-5, "d k")


def test_hfe_block_hip_helpers_vs_module_path():
    """HFEBlock with the HIP helpers (Gram-based matching + attention, LayerNorm2d, depth-wise conv) vs the
    same block evaluated with the plain PyTorch ops on the GPU."""
    torch.manual_seed(3)
    blk = arch.HFEBlock(32, match_factor=1, ffn_expansion_factor=1).eval().to(DEV)
    x = torch.randn(1, 32, 40, 56, device=DEV)
    per = torch.randn(1, 32, 40, 56, device=DEV)
    with torch.no_grad():
        fast = blk(x, per)
        prev = oracle_backend.set_ops_backend(type("Plain", (), {})())        # a backend without any helper
        try:
            ref = blk(x, per)
        finally:
            oracle_backend.set_ops_backend(prev)
    assert_close(fast, ref, TOL, "HFEBlock")


# ------------------------------------------------------------------------------------------------
# network level
# ------------------------------------------------------------------------------------------------
def test_tiny_model_golden(golden):
    g = golden("model_tiny")
    net = wm.WaveMamba(in_chn=3, wf=8, n_l_blocks=[1, 1, 2], n_h_blocks=[1, 1, 1], ffn_scale=2.0).eval()
    net.load_state_dict({k[2:]: v for k, v in g.items() if k.startswith("p.")}, strict=True)
    net = net.to(DEV)
    with torch.no_grad():
        y = net(g["x"].to(DEV))
    assert_close(y, g["y"], TOL, "tiny model")


@pytest.mark.parametrize("tag,hw", [("32x64", (32, 64)), ("128x128", (128, 128)), ("256x256", (256, 256))])
def test_shipped_config_golden(golden, tag, hw):
    torch.manual_seed(0)
    net = wm.WaveMamba(in_chn=3, wf=32, n_l_blocks=[1, 2, 4], n_h_blocks=[1, 1, 2], ffn_scale=2.0).eval().to(DEV)
    x = torch.rand(1, 3, *hw, generator=gen(1234))
    with torch.no_grad():
        y = net.restoration_network(x.to(DEV))
    want = golden("model_shipped")[f"y_{tag}"]
    assert_close(y, want, TOL, f"shipped {tag}")
    # PSNR match vs the reference output after the reference's uint8 quantisation
    # (img_util.py:67-94, comput_psnr_ssim.py:434-438), against a seeded synthetic target
    tgt = torch.rand(1, 3, *hw, generator=gen(4321))

    def psnr(a, b):
        qa = (a.clamp(0, 1) * 255).round()
        qb = (b.clamp(0, 1) * 255).round()
        return float(20 * torch.log10(255.0 / (qa - qb).pow(2).mean().sqrt()))
    assert abs(psnr(y.cpu(), tgt) - psnr(want, tgt)) <= 1e-3


# ------------------------------------------------------------------------------------------------
# depth-wise 3x3 conv (+bias, +SiLU): SS2D.conv2d / ffn.conv2 (reference :346-355, :487, :220)
# floating-point kernel of a standard op -> PyTorch fp32 CPU conv is the reference
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(1, 64, 32, 48), (2, 5, 7, 9), (1, 3, 1, 1), (1, 8, 130, 260),
                                   (1, 2, 33, 1030), (1, 96, 17, 64),
                                   (8, 64, 64, 64), (3, 7, 37, 64), (1, 5, 16, 8), (2, 3, 70, 128), (1, 9, 20, 100), (1, 1, 3, 4)])
@pytest.mark.parametrize("act", ["none", "silu"])
def test_dwconv3x3_vs_torch(shape, act):
    import torch.nn.functional as F
    B, C, H, W = shape
    gg = gen(B * 1000 + W)
    x = torch.randn(*shape, generator=gg)
    w = torch.randn(C, 1, 3, 3, generator=gg) * 0.3
    b = torch.randn(C, generator=gg)
    ref = F.conv2d(x, w, b, stride=1, padding=1, groups=C)
    if act == "silu":
        ref = F.silu(ref)
    got = wm.ops.dwconv3x3(x.to(DEV), w.to(DEV), b.to(DEV), act)
    assert_close(got, ref, 1e-5, f"dwconv {shape} {act}")
    got_nb = wm.ops.dwconv3x3(x.to(DEV), w.to(DEV), None, "none")
    assert_close(got_nb, F.conv2d(x, w, None, padding=1, groups=C), 1e-5, "dwconv no bias")


# ------------------------------------------------------------------------------------------------
# dense 3x3 / 1x1 conv on the bf16 matrix cores with a two-term operand split, fused with its element-wise
# neighbours: PAConv.k2/.k3/.k4 on cat([x, gather(p, idx)]), qkv / project_in / project_out (+ residual), l_conv on
# cat([LL, x_d]), h_out_conv, conv_01, last (+ img), ps_down* (reference :690-697, :666/:713, :733-797, :966/:975,
# :993/:1006, :1015-1037).  Floating-point kernel of standard ops -> the fp64 PyTorch CPU composition is the
# reference; bar: 2e-5 relative (three bf16 products per term: <= 3 * 2^-18 per product; measured ~4e-6)
# ------------------------------------------------------------------------------------------------
@pytest.fixture(params=["first_gen", "wave_specialised"])
def conv3x3_impl(request):
    """Run the test once per 3x3 kernel (the automatic choice would send these small shapes to the first-generation
    one only)."""
    wm.ops.conv2d_select(wm.ops.CONV3X3_FIRST_GEN if request.param == "first_gen" else wm.ops.CONV3X3_WAVE_SPECIALISED)
    yield request.param
    wm.ops.conv2d_select(wm.ops.CONV3X3_AUTO)


@pytest.mark.parametrize("ks", [3, 1])
@pytest.mark.parametrize("B,Ca,Cb,Cout,H,W,bias", [
    (1, 64, 0, 64, 70, 50, True), (2, 64, 0, 32, 33, 65, False), (1, 32, 32, 32, 40, 96, True),
    (1, 32, 0, 96, 16, 32, True), (1, 3, 0, 32, 40, 64, True), (2, 32, 0, 3, 64, 96, True),
    (1, 16, 8, 40, 5, 7, False), (1, 64, 0, 64, 1, 1, True), (1, 48, 0, 64, 31, 33, False),
    (1, 64, 0, 64, 160, 256, False), (1, 12, 0, 32, 24, 40, True), (1, 192, 0, 32, 9, 17, True)])
def test_conv2d_vs_torch(conv3x3_impl, ks, B, Ca, Cb, Cout, H, W, bias):
    import torch.nn.functional as F
    gg = gen(Ca * 100 + Cout + H + ks)
    xa = torch.randn(B, Ca, H, W, generator=gg)
    xb = torch.randn(B, Cb, H, W, generator=gg) if Cb else None
    w = torch.randn(Cout, Ca + Cb, ks, ks, generator=gg) / (ks * (Ca + Cb) ** 0.5)
    b = torch.randn(Cout, generator=gg) if bias else None
    xin = xa if xb is None else torch.cat([xa, xb], 1)
    ref = F.conv2d(xin.double(), w.double(), None if b is None else b.double(), padding=ks // 2).float()
    got = wm.ops.conv2d(*cu(xa, w, b, xb))
    assert_close(got, ref, 2e-5, f"conv2d ks={ks} {(B, Ca, Cb, Cout, H, W)}")


@pytest.mark.parametrize("two_streams", [False, True])
def test_forward_ignores_allocator_pool_contents(two_streams):
    """No kernel of the inference forward consumes memory it does not own or has not written (uninitialised `torch.empty` outputs,
    out-of-bounds tile reads): with every stream's allocator pool filled with NaN / 1e30 / 0 before the forward (blocks of many
    sizes allocated, filled, freed) the output stays bit-equal - ragged map sizes included (49-, 98-, 25-column levels)."""
    net = wm.WaveMamba(in_chn=3, wf=32, n_l_blocks=[1, 2, 4], n_h_blocks=[1, 1, 2], ffn_scale=2.0).eval().to(DEV)
    unet = net.restoration_network
    xs = [torch.rand(1, 3, 264, 392, generator=gen(41)).to(DEV), torch.rand(2, 3, 136, 200, generator=gen(42)).to(DEV)]
    sizes = [2 ** k for k in range(9, 25)] + [3 * 2 ** k for k in range(9, 23)]            # 512 B .. 16 MB

    def poison(value, streams):
        for st in streams:
            with torch.cuda.stream(st):
                held = []
                for n in sizes + sizes:
                    t = torch.empty(n // 4, dtype=torch.float32, device=DEV); t.fill_(value); held.append(t)
                del held
        torch.cuda.synchronize()

    old = unet.two_streams
    try:
        unet.two_streams = two_streams
        with torch.no_grad():
            clean = [unet(x).clone() for x in xs]
            torch.cuda.synchronize()
            streams = [torch.cuda.current_stream(DEV)] + (list(arch._side_streams(xs[0], 3)) if two_streams else [])
            for value in (float("nan"), 1e30, 0.0):
                for x, c in zip(xs, clean):
                    poison(value, streams)
                    o = unet(x)
                    torch.cuda.synchronize()
                    assert torch.equal(o, c), f"poison {value}, input {tuple(x.shape)}: the forward read memory it had not written"
    finally:
        unet.two_streams = old


@pytest.mark.parametrize("ks", [3, 1])
@pytest.mark.parametrize("B,Ca,Csrc,Cb,Cout,H,W", [(1, 32, 32, 32, 64, 24, 40), (2, 32, 32, 32, 64, 9, 33),
                                                  (2, 16, 24, 8, 32, 17, 31)])
def test_conv2d_gather_gate_residual(conv3x3_impl, ks, B, Ca, Csrc, Cb, Cout, H, W):
    """cat([x, gather(p, idx)]) as operand, `* sigmoid(gate)` and `+ residual` in the epilogue: the PAConv /
    HFEBlock composition of the reference (:666, :694-697, :713, :849-853)."""
    import torch.nn.functional as F
    gg = gen(B * 31 + Cb + ks)
    x = torch.randn(B, Ca, H, W, generator=gg)
    p = torch.randn(B, Csrc, H, W, generator=gg)
    idx = torch.randint(0, Csrc, (B, Cb), generator=gg)
    w = torch.randn(Cout, Ca + Cb, ks, ks, generator=gg) / (ks * (Ca + Cb) ** 0.5)
    b = torch.randn(Cout, generator=gg)
    gate = torch.randn(B, Cout, H, W, generator=gg)
    res = torch.randn(B, Cout, H, W, generator=gg)
    xin = torch.cat([x, torch.gather(p, 1, idx[:, :, None, None].expand(-1, -1, H, W))], 1)
    conv = F.conv2d(xin.double(), w.double(), b.double(), padding=ks // 2)
    xd, wd, bd, pd, idxd, gd, rd = cu(x, w, b, p, idx, gate, res)
    assert_close(wm.ops.conv2d(xd, wd, bd, pd, idxd), conv.float(), 2e-5, "conv2d gather")
    assert_close(wm.ops.conv2d(xd, wd, bd, pd, idxd, gate=gd), (conv * torch.sigmoid(gate.double())).float(), 2e-5,
                 "conv2d gather + gate")
    assert_close(wm.ops.conv2d(xd, wd, bd, pd, idxd, gate=gd, residual=rd),
                 (conv * torch.sigmoid(gate.double()) + res.double()).float(), 2e-5, "conv2d gather + gate + residual")
    assert_close(wm.ops.conv2d(xd, wd, None, pd, idxd, residual=rd),
                 (F.conv2d(xin.double(), w.double(), None, padding=ks // 2) + res.double()).float(), 2e-5,
                 "conv2d gather + residual, no bias")


def test_conv2d_weight_update_invalidates_prepared_copy():
    import torch.nn.functional as F
    gg = gen(5)
    x = torch.randn(1, 32, 24, 40, generator=gg).to(DEV)
    conv = torch.nn.Conv2d(32, 32, 3, 1, 1).to(DEV)
    y0 = wm.ops.conv2d(x, conv.weight, conv.bias)
    with torch.no_grad():
        conv.weight.mul_(2.0)                                   # in-place update, as an optimizer step does
    y1 = wm.ops.conv2d(x, conv.weight, conv.bias)
    ref = F.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=1).float()
    assert_close(y1, ref.cpu(), 2e-5, "conv2d after in-place weight update")
    assert not torch.allclose(y0, y1)
    # a write through .data does not bump the version counter: the explicit hook (called by WaveMamba.train() / .eval() /
    # ._apply() and trainer.load_network) drops the stale copy
    conv.weight.data.mul_(0.5)
    wm.ops.conv2d_cache_clear()
    assert_close(wm.ops.conv2d(x, conv.weight, conv.bias), y0.cpu(), 2e-5, "conv2d after .data update + cache clear")
    # a prepared copy built on one stream and first used from another: the using stream waits for the preparation
    wm.ops.conv2d_cache_clear()
    side = torch.cuda.Stream(DEV)
    torch.cuda.synchronize()
    ya = wm.ops.conv2d(x, conv.weight, conv.bias)              # builds the copy on the current stream
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        yb = wm.ops.conv2d(x, conv.weight, conv.bias)          # same copy, other stream
    torch.cuda.synchronize()
    assert torch.equal(ya, yb)


def test_fused_adamw_step_then_inference_uses_fresh_weights():
    """a fused multi-tensor AdamW step followed by a no_grad forward must see the updated weights in the
    inference convolutions' prepared copies."""
    torch.manual_seed(0)
    net = wm.WaveMamba(in_chn=3, wf=8, n_l_blocks=[1, 1, 1], n_h_blocks=[1, 1, 1], ffn_scale=2.0).to(DEV)
    x = torch.rand(1, 3, 32, 48, generator=gen(2)).to(DEV)
    net.eval()
    with torch.no_grad():
        y0 = net(x)
    net.train()
    opt = wm.trainer.make_optimizer(net)
    wm.trainer.train_step(net, opt, x, torch.rand(1, 3, 32, 48, generator=gen(3)).to(DEV))
    net.eval()
    with torch.no_grad():
        y1 = net(x)
        saved = wm.ops.conv2d_supported
        wm.ops.conv2d_supported = lambda *a, **k: False        # the same weights through PyTorch's convolutions
        try:
            y_ref = net(x)
        finally:
            wm.ops.conv2d_supported = saved
    assert not torch.allclose(y0, y1)
    assert_close(y1, y_ref, 1e-4, "forward after an optimizer step")


# ------------------------------------------------------------------------------------------------
# small-tensor HFE kernels (csrc/hfe.hip.h) against the PyTorch composition of the reference
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,C,L", [(1, 32, 4096), (2, 32, 777), (3, 8, 64)])
def test_match_index_vs_cdist_topk(B, C, L):
    """reference :659-666 with every channel kept: nearest candidate channel under the L2 distance."""
    gg = gen(B * 7 + C)
    x = torch.randn(B, C, L, generator=gg)
    p = torch.randn(B, C, L, generator=gg)
    p[:, 3] = x[:, 5] + 1e-3 * torch.randn(B, L, generator=gg)      # an unambiguous nearest pair
    ref = torch.cdist(x.double(), p.double()).topk(k=1, largest=False)[1].squeeze(-1)
    G, nx, ny = wm.ops.gram(*cu(x, p))
    got = wm.ops.match_index(G, nx, ny).cpu().long()
    assert torch.equal(got, ref)
    assert got[0, 5] == 3


@pytest.mark.parametrize("B,C,heads,L", [(1, 32, 1, 2000), (2, 32, 2, 640), (1, 64, 4, 333)])
def test_attn_fold_vs_reference_composition(B, C, heads, L):
    """project_out(softmax(normalize(q) @ normalize(k)^T * temperature) @ v), reference :787-797."""
    import torch.nn.functional as F
    gg = gen(C + heads)
    ch = C // heads
    q = torch.randn(B * heads, ch, L, generator=gg)
    k = torch.randn(B * heads, ch, L, generator=gg)
    v = torch.randn(B, C, L, generator=gg)
    temp = torch.rand(heads, generator=gg) + 0.5
    wpo = torch.randn(C, C, 1, 1, generator=gg) / C ** 0.5
    qn = F.normalize(q.double().reshape(B, heads, ch, L), dim=-1)
    kn = F.normalize(k.double().reshape(B, heads, ch, L), dim=-1)
    attn = ((qn @ kn.transpose(-2, -1)) * temp.double().view(1, heads, 1, 1)).softmax(dim=-1)
    ref = torch.einsum("oc,bcl->bol", wpo.double().view(C, C), (attn @ v.double().reshape(B, heads, ch, L)).reshape(B, C, L))
    G, nq, nk = wm.ops.gram(*cu(q, k))
    wf = wm.ops.attn_fold(G, nq, nk, temp.to(DEV), wpo.to(DEV), B, heads)
    got = torch.einsum("boc,bcl->bol", wf.double().cpu(), v.double())
    assert_close(got.float(), ref.float(), 2e-5, "attention folded into project_out")


@pytest.mark.parametrize("B,C,H,W", [(1, 32, 40, 72), (2, 32, 17, 23), (1, 16, 8, 8)])
def test_skff_vs_module_path(B, C, H, W):
    torch.manual_seed(11)
    m = arch.SKFF(C, height=3, reduction=8).eval()
    with torch.no_grad():
        m.conv_du[1].weight.fill_(0.2)
    feats = [torch.randn(B, C, H, W, generator=gen(i + 3)) for i in range(3)]
    with torch.no_grad():
        ref = m.double()([f.double() for f in feats]).float()      # CPU: no ops backend kernel applies -> PyTorch path
        got = m.float().to(DEV)([f.to(DEV) for f in feats])
    assert_close(got, ref, 1e-5, "SKFF")


@pytest.mark.parametrize("shape", [(1, 32, 33, 47), (2, 5, 7, 9)])
def test_dwconv3x3_gelu(shape):
    import torch.nn.functional as F
    B, C, H, W = shape
    gg = gen(W)
    x = torch.randn(*shape, generator=gg)
    w = torch.randn(C, 1, 3, 3, generator=gg) * 0.3
    b = torch.randn(C, generator=gg)
    ref = F.gelu(F.conv2d(x.double(), w.double(), b.double(), padding=1, groups=C)).float()
    assert_close(wm.ops.dwconv3x3(*cu(x, w, b), "gelu"), ref, 1e-5, "dwconv + gelu")


# ------------------------------------------------------------------------------------------------
# fused SS2D core backward (wm_ss2d_core_bwd) against the reference's autograd of forward_core
# ------------------------------------------------------------------------------------------------
def core_grads_f64(x, Wx, Wdt, bias, A_logs, Ds, dys):
    """SS2D.forward_core (:446-478) with the sequential scan definition, all in float64, through autograd: the truth for
    the fused backward.  dys: four (B, D, L) tensors in the reference's return order, or one (merged)."""
    leaves = [t.detach().cpu().double().requires_grad_(True) for t in (x, Wx, Wdt, bias, A_logs, Ds)]
    x_, Wx_, Wdt_, b_, Al_, Ds_ = leaves
    B, D, H, W = x_.shape
    L, K = H * W, 4
    R, N = Wdt_.shape[2], Al_.shape[1]
    xs = torch.stack([x_.view(B, -1, L), x_.transpose(2, 3).contiguous().view(B, -1, L)], dim=1).view(B, 2, -1, L)
    xs = torch.cat([xs, torch.flip(xs, dims=[-1])], dim=1)
    x_dbl = torch.einsum("b k d l, k c d -> b k c l", xs, Wx_)
    dts, Bs, Cs = torch.split(x_dbl, [R, N, N], dim=2)
    dts = torch.einsum("b k r l, k d r -> b k d l", dts, Wdt_)
    u = xs.reshape(B, K * D, L)
    dt = F.softplus(dts.reshape(B, K * D, L) + b_.reshape(1, -1, 1))
    A = -torch.exp(Al_)
    Bf = Bs.repeat_interleave(D, dim=1)                 # (B, K D, N, L)
    Cf = Cs.repeat_interleave(D, dim=1)
    h = torch.zeros(B, K * D, N, dtype=torch.float64)
    ys = []
    for dt_t, du_t, B_t, C_t in zip(dt.unbind(2), (dt * u).unbind(2), Bf.unbind(3), Cf.unbind(3)):
        h = torch.exp(dt_t[..., None] * A.view(1, K * D, N)) * h + du_t[..., None] * B_t
        ys.append((h * C_t).sum(-1))
    out = (torch.stack(ys, 2) + u * Ds_.view(1, -1, 1)).view(B, K, -1, L)
    inv = torch.flip(out[:, 2:4], dims=[-1]).view(B, 2, -1, L)
    wh = out[:, 1].view(B, -1, W, H).transpose(2, 3).contiguous().view(B, -1, L)
    invwh = inv[:, 1].view(B, -1, W, H).transpose(2, 3).contiguous().view(B, -1, L)
    outs = (out[:, 0], inv[:, 0], wh, invwh)
    dys = [d.detach().cpu().double() for d in dys]
    if len(dys) == 1:
        return torch.autograd.grad(sum(outs), leaves, dys[0])
    return torch.autograd.grad(outs, leaves, dys)


def assert_vs_truth(got, ref, truth, what):
    """err(got, truth) <= max(1e-4, 2 err(ref, truth)): `ref` is the fp32 result the build is compared with."""
    e_got, e_ref = max(rel_err(got, truth)), max(rel_err(ref, truth))
    assert e_got <= truth_bar(e_ref), f"{what}: {e_got:.3e} vs float64 truth (fp32 reference: {e_ref:.3e})"


# ------------------------------------------------------------------------------------------------
# fused SS2D core backward (wm_ss2d_core_bwd) against the reference's autograd of forward_core
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["s16", "sq16", "c32", "d8"])       # c32: d_state 32 (BASELINE config 5's block)
def test_ss2d_core_backward_golden(golden, tag):
    """Gradients of the reference's real SS2D.forward_core (tests/golden/make_golden.py) for four random dys: within
    1e-4 of the reference's fp32 autograd, or - the parameter gradients are sums over every position - judged with it
    against the float64 evaluation of the same formula."""
    g = golden("scan")
    names = ("core_x", "x_proj_weight", "dt_projs_weight", "dt_projs_bias", "A_logs", "Ds")
    args = [t.clone().requires_grad_(True) for t in cu(*[g[f"{tag}_{n}"] for n in names])]
    ys = wm.ops.ss2d_core(*args)
    dys = cu(*[g[f"{tag}_core_dy{i}"] for i in range(4)])
    grads = torch.autograd.grad(ys, args, dys)
    refs = ("core_dx", "core_dx_proj_weight", "core_ddt_projs_weight", "core_ddt_projs_bias", "core_dA_logs", "core_dDs")
    truth = core_grads_f64(*args, dys)
    for got, rn, tr in zip(grads, refs, truth):
        assert_vs_truth(got, g[f"{tag}_{rn}"], tr, f"{tag} {rn}")


@pytest.mark.parametrize("B,D,H,W,N,R", [(1, 64, 12, 20, 16, 2), (2, 32, 7, 9, 16, 2), (1, 16, 33, 5, 8, 1),
                                         (1, 64, 40, 48, 16, 4), (1, 64, 24, 20, 32, 2), (2, 64, 9, 8, 24, 3)])
@pytest.mark.parametrize("merged", [True, False])
def test_ss2d_core_backward_vs_unfused_autograd(B, D, H, W, N, R, merged):
    """Fused backward against PyTorch autograd through the direction glue + the HIP op-boundary scan (itself checked
    against the reference goldens and the oracle above), both judged against the float64 evaluation."""
    x, Wx, Wdt, bias, A_logs, Ds = [t.to(DEV).requires_grad_(True) for t in random_core_case(B, D, H, W, N, R, seed=H + W)]
    L = H * W

    def unfused():
        xs = torch.stack([x.view(B, -1, L), x.transpose(2, 3).contiguous().view(B, -1, L)], dim=1).view(B, 2, -1, L)
        xs = torch.cat([xs, torch.flip(xs, dims=[-1])], dim=1)
        x_dbl = torch.einsum("b k d l, k c d -> b k c l", xs, Wx)
        dts, Bs, Cs = torch.split(x_dbl, [R, N, N], dim=2)
        dts = torch.einsum("b k r l, k d r -> b k d l", dts, Wdt)
        out = wm.ops.selective_scan_fn(xs.reshape(B, -1, L), dts.reshape(B, -1, L), -torch.exp(A_logs), Bs.contiguous(),
                                       Cs.contiguous(), Ds, None, bias.reshape(-1), True).view(B, 4, -1, L)
        inv = torch.flip(out[:, 2:4], dims=[-1]).view(B, 2, -1, L)
        wh = out[:, 1].view(B, -1, W, H).transpose(2, 3).contiguous().view(B, -1, L)
        invwh = inv[:, 1].view(B, -1, W, H).transpose(2, 3).contiguous().view(B, -1, L)
        return out[:, 0], inv[:, 0], wh, invwh

    params = [x, Wx, Wdt, bias, A_logs, Ds]
    gg = torch.Generator(device=DEV).manual_seed(5)
    if merged:
        dys = [torch.randn(B, D, L, device=DEV, generator=gg)]
        ref = torch.autograd.grad(sum(unfused()), params, dys[0])
        got = torch.autograd.grad(wm.ops.ss2d_core(*params, merged=True), params, dys[0])
    else:
        dys = [torch.randn(B, D, L, device=DEV, generator=gg) for _ in range(4)]
        ref = torch.autograd.grad(unfused(), params, dys)
        got = torch.autograd.grad(wm.ops.ss2d_core(*params), params, dys)
    truth = core_grads_f64(*params, dys)
    for a, b, tr, nm in zip(got, ref, truth, ("dx", "dWx", "dWdt", "dbias", "dA_logs", "dDs")):
        assert_vs_truth(a, b, tr, f"fused core bwd {nm} {(B, D, H, W, N, R)} merged={merged}")


def core_truth_subset_f64(x, Wx, Wdt, bias, A_logs, Ds, dy, chans):
    """SS2D.forward_core's backward (:446-478 under autograd) in FLOAT64 ON THE GPU with the
    SEQUENTIAL recurrence, for the channels `chans` of every direction and a merged output gradient dy (B, D, L): the truth
    at map sizes where autograd through a float64 Python loop cannot run (L = 65,536 at BASELINE config 3's level 1).
    Operands are formed the way the reference forms them (all 64 channels enter x_dbl), the recurrence runs one step per
    launch (h_t = a_t h_{t-1} + b_t forward, g_t = C_t dy_t + a_{t+1} g_{t+1} backward), everything else is vectorised.
    Returns per-channel gradients (4, S, ...): dA_logs, dDs, d dt_projs_bias, d dt_projs_weight."""
    x, Wx, Wdt, bias, A_logs, Ds, dy = [t.detach().double() for t in (x, Wx, Wdt, bias, A_logs, Ds, dy)]
    B, D, H, W = x.shape
    L, N, R, S = H * W, A_logs.shape[1], Wdt.shape[2], len(chans)
    ci = torch.tensor(chans, device=x.device)

    def seq(t, k):          # (B, C, H, W) map -> (B, C, L) in direction k's scan order (:451-452)
        v = t.reshape(B, -1, L) if k % 2 == 0 else t.transpose(2, 3).reshape(B, -1, L)
        return torch.flip(v, dims=[-1]) if k >= 2 else v

    dyk_map = dy.view(B, D, H, W)
    a_l, b_l, c_l, pack = [], [], [], []
    for k in range(4):
        xs = seq(x, k)
        x_dbl = torch.einsum("bdl,cd->bcl", xs, Wx[k])
        dt_r, Bk, Ck = x_dbl[:, :R], x_dbl[:, R:R + N], x_dbl[:, R + N:]
        delta = torch.einsum("brl,dr->bdl", dt_r, Wdt[k, ci]) + bias[k, ci].view(1, S, 1)
        dt = F.softplus(delta)
        u = xs[:, ci]
        dyk = seq(dyk_map, k)[:, ci]
        A = -torch.exp(A_logs[k * D + ci])                                            # (S, N)
        tm = lambda t: t.permute(2, 0, 1).contiguous()                               # (B, S, L) -> (L, B, S)
        a_l.append(torch.exp(tm(dt)[..., None] * A))                                  # (L, B, S, N)
        b_l.append(tm(dt * u)[..., None] * Bk.permute(2, 0, 1)[:, :, None, :])
        c_l.append(tm(dyk)[..., None] * Ck.permute(2, 0, 1)[:, :, None, :])
        pack.append((dt, u, dyk, delta, A, Bk, dt_r))
    a_all, b_all, c_all = (torch.cat(v, dim=2) for v in (a_l, b_l, c_l))              # (L, B, 4 S, N)
    del a_l, b_l, c_l
    h_all = torch.empty_like(b_all)
    h_all[0] = b_all[0]
    for t in range(1, L):
        torch.addcmul(b_all[t], a_all[t], h_all[t - 1], out=h_all[t])
    g_all = b_all                                                                     # (b is consumed: reuse its storage)
    g_all[L - 1] = c_all[L - 1]
    for t in range(L - 2, -1, -1):
        torch.addcmul(c_all[t], a_all[t + 1], g_all[t + 1], out=g_all[t])
    hprev = torch.cat([torch.zeros_like(h_all[:1]), h_all[:-1]], dim=0)
    gah = g_all * a_all * hprev                                                       # g_t a_t h_{t-1}
    del hprev, h_all, a_all, c_all
    out = {"dA_logs": [], "dDs": [], "dbias": [], "dWdt": []}
    for k, (dt, u, dyk, delta, A, Bk, dt_r) in enumerate(pack):
        gk, gahk = g_all[:, :, k * S:(k + 1) * S], gah[:, :, k * S:(k + 1) * S]       # (L, B, S, N)
        dtm = dt.permute(2, 0, 1)
        dA = (gahk * dtm[..., None]).sum(dim=(0, 1))
        gB = (gk * Bk.permute(2, 0, 1)[:, :, None, :]).sum(-1)                        # <g_t, B_t>   (L, B, S)
        ddt = (gahk * A).sum(-1) + u.permute(2, 0, 1) * gB
        ddelta = ddt * torch.sigmoid(delta.permute(2, 0, 1))
        out["dA_logs"].append(dA * A)
        out["dDs"].append((dyk * u).sum(dim=(0, 2)))
        out["dbias"].append(ddelta.sum(dim=(0, 1)))
        out["dWdt"].append(torch.einsum("lbs,brl->sr", ddelta, dt_r))
    return {n: torch.stack(v) for n, v in out.items()}


@pytest.mark.parametrize("B,D,H,W", [(8, 64, 256, 256), (8, 64, 128, 128), (8, 64, 64, 64), (1, 64, 544, 960)])
def test_ss2d_core_backward_at_training_sizes(B, D, H, W):
    """wm_ss2d_core_bwd at the map sizes BASELINE config 3 times it at (batch 8, levels 1-3 of 512 x 512 crops: L = 65,536
    = 256 blocks of 256 steps per direction, two-level carries, batch strides, many projgrad partials, real transposes) and at
    one UHD level-2 map.  (1) Every gradient against autograd through the UNFUSED path (torch direction glue + einsums + the
    op-boundary HIP scan with its own backward): dx / dWx <= 1e-4.  (2) The per-channel parameter gradients (dA_logs, dDs,
    d dt_projs_bias, d dt_projs_weight) of four channels of EVERY direction against the float64 sequential recurrence on
    the same operands (core_truth_subset_f64), judged like every gradient test: err <= max(1e-4, 2 err(unfused fp32))."""
    N, R = 16, 2
    x, Wx, Wdt, bias, A_logs, Ds = [t.to(DEV).requires_grad_(True) for t in random_core_case(B, D, H, W, N, R, seed=H + W + B)]
    L = H * W
    params = [x, Wx, Wdt, bias, A_logs, Ds]
    dy = torch.randn(B, D, L, device=DEV, generator=torch.Generator(device=DEV).manual_seed(7))

    def unfused():
        xs = torch.stack([x.view(B, -1, L), x.transpose(2, 3).contiguous().view(B, -1, L)], dim=1).view(B, 2, -1, L)
        xs = torch.cat([xs, torch.flip(xs, dims=[-1])], dim=1)
        x_dbl = torch.einsum("b k d l, k c d -> b k c l", xs, Wx)
        dts, Bs, Cs = torch.split(x_dbl, [R, N, N], dim=2)
        dts = torch.einsum("b k r l, k d r -> b k d l", dts, Wdt)
        out = wm.ops.selective_scan_fn(xs.reshape(B, -1, L), dts.reshape(B, -1, L), -torch.exp(A_logs), Bs.contiguous(),
                                       Cs.contiguous(), Ds, None, bias.reshape(-1), True).view(B, 4, -1, L)
        inv = torch.flip(out[:, 2:4], dims=[-1]).view(B, 2, -1, L)
        wh = out[:, 1].view(B, -1, W, H).transpose(2, 3).contiguous().view(B, -1, L)
        invwh = inv[:, 1].view(B, -1, W, H).transpose(2, 3).contiguous().view(B, -1, L)
        return out[:, 0] + inv[:, 0] + wh + invwh

    y_ref = unfused()
    ref = torch.autograd.grad(y_ref, params, dy)
    y_got = wm.ops.ss2d_core(*params, merged=True)
    assert_close(y_got, y_ref, TOL, f"fused core forward {(B, D, H, W)}")
    del y_ref
    got = torch.autograd.grad(y_got, params, dy)
    names = ("dx", "dWx", "dWdt", "dbias", "dA_logs", "dDs")
    for a, b, nm in zip(got, ref, names):
        assert torch.isfinite(a).all(), nm
        assert_close(a, b, TOL, f"fused core bwd {nm} {(B, D, H, W)} vs unfused autograd")
    chans = [0, 21, 42, 63]
    truth = core_truth_subset_f64(x, Wx, Wdt, bias, A_logs, Ds, dy, chans)
    ci = torch.tensor(chans, device=DEV)
    sel = {"dA_logs": lambda g: g.view(4, D, N)[:, ci], "dDs": lambda g: g.view(4, D)[:, ci],
           "dbias": lambda g: g.view(4, D)[:, ci], "dWdt": lambda g: g.view(4, D, R)[:, ci]}
    worst = {}
    for nm, pick in sel.items():
        i = names.index(nm)
        worst[nm] = (max(rel_err(pick(got[i]), truth[nm])), max(rel_err(pick(ref[i]), truth[nm])))
        assert_vs_truth(pick(got[i]), pick(ref[i]), truth[nm], f"fused core bwd {nm} {(B, D, H, W)} channels {chans}")
    print(f"core bwd {(B, D, H, W)} vs float64 recurrence (fused, unfused): " +
          ", ".join(f"{k} {v[0]:.1e}/{v[1]:.1e}" for k, v in worst.items()))


@pytest.mark.parametrize("B,D,H,W", [(8, 64, 256, 256), (8, 64, 64, 64)])
def test_ss2d_core_backward_bit_reproducible_stress(B, D, H, W):
    """round 4 saw `dx` rows of the accumulating (mirrored) launches of wm_ss2d_core_bwd zeroed in lanes 48..63 now and
    then and attributed it to a `v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0]` the compiler had fused.  Round 5 reproduced that
    instruction form failing on its own (tools/ubench_pk_coexec.hip: a packed-fp32 add with one source SWAPPED returns zero halves
    in lanes 48..63 while LDS-fed MFMAs share the SIMD - here the kernel's own projection waves): the diagnosis holds, the form
    is linted out of the library (tools/lint_packed_f32.py).  This is the stress test asked for: 300 backward calls at the
    config-3 map sizes, every gradient bit-equal to the first call's, with and without a 3x3 matrix-core convolution running
    on a second stream."""
    N, R = 16, 2
    x, Wx, Wdt, bias, A_logs, Ds = [t.to(DEV).requires_grad_(True) for t in random_core_case(B, D, H, W, N, R, seed=3 * H + B)]
    params = [x, Wx, Wdt, bias, A_logs, Ds]
    dy = torch.randn(B, D, H * W, device=DEV, generator=torch.Generator(device=DEV).manual_seed(9))
    xa = torch.randn(1, 64, 544, 960, generator=gen(12)).to(DEV)
    w3 = (torch.randn(64, 64, 3, 3, generator=gen(13)) / 24).to(DEV)
    side = torch.cuda.Stream(DEV)

    def grads():
        y = wm.ops.ss2d_core(*params, merged=True)
        return torch.autograd.grad(y, params, dy)
    ref = [g.clone() for g in grads()]
    torch.cuda.synchronize()
    names = ("dx", "dWx", "dWdt", "dbias", "dA_logs", "dDs")
    for concurrent in (False, True):
        cnt = [torch.zeros((), dtype=torch.int64, device=DEV) for _ in names]
        keep = []
        for _ in range(150):
            if concurrent:
                with torch.cuda.stream(side), torch.no_grad():
                    keep.append(wm.ops.conv2d(xa, w3))
                    if len(keep) > 6:
                        keep.pop(0)
            for c, g, r in zip(cnt, grads(), ref):
                c += (g != r).sum()
        torch.cuda.synchronize()
        bad = {n: int(c) for n, c in zip(names, cnt) if int(c)}
        assert not bad, f"{(B, D, H, W)} concurrent conv3x3 {concurrent}: differing gradient elements over 150 backward calls: {bad}"


def test_trainable_lfss_block_d_state_32(golden):
    """BASELINE config 5's block in training: LFSSBlock(32, d_state=32) forward + backward on the HIP training path
    (fused core forward wm_ss2d_core_fwd, backward wm_ss2d_core_bwd at N = 32) against the REFERENCE's autograd
    (tests/golden/lfss_block_n32.npz: output, dx, every parameter gradient, in fp32 and in float64)."""
    g = golden("lfss_block_n32")
    blk = arch.LFSSBlock(32, d_state=32, expand=2.0).train()
    blk.load_state_dict({k[2:]: v for k, v in g.items() if k.startswith("p.")}, strict=True)
    blk = blk.to(DEV)
    x = g["x"].to(DEV).requires_grad_(True)
    assert wm.ops.ss2d_core_bwd_supported(64, 32, 2)
    y = blk(x, [16, 12])
    assert_close(y, g["y"], TOL, "LFSSBlock(d_state=32) output")
    params = dict(blk.named_parameters())
    grads = torch.autograd.grad(y, [x] + list(params.values()), g["dy"].to(DEV))
    assert_vs_truth(grads[0], g["dx"], g["dx_f64"], "dx")
    for k, got in zip(params, grads[1:]):
        assert_vs_truth(got, g["g." + k], g["t." + k], f"gradient of {k}")


def test_batched_forward_equals_per_image_forward():
    """Batch strides of every fused path (gathered conv operands, per-image folded attention weights, SKFF, merged
    scans): a 2-image batch must reproduce the two single-image forwards."""
    torch.manual_seed(0)
    net = wm.WaveMamba(in_chn=3, wf=32, n_l_blocks=[1, 2, 4], n_h_blocks=[1, 1, 2], ffn_scale=2.0).eval().to(DEV)
    x = torch.rand(2, 3, 64, 96, generator=gen(9)).to(DEV)
    with torch.no_grad():
        both = net.restoration_network(x)
        one = torch.cat([net.restoration_network(x[i:i + 1]) for i in range(2)], 0)
    assert_close(both, one.cpu(), 1e-5, "batch of 2 vs two single images")


def test_multi_stream_forward_is_the_single_stream_forward():
    """UNet.forward issues each DownFRG's high-frequency branch on a side stream (inference).  Same kernels on the same
    inputs: the output must equal the single-stream order bit for bit - also back to back without a host
    synchronisation in between (a missing cross-stream ordering or a buffer the allocator handed out too early shows as a
    difference), from a non-default current stream, and with a batch."""
    net = wm.WaveMamba(in_chn=3, wf=32, n_l_blocks=[1, 2, 4], n_h_blocks=[1, 1, 2], ffn_scale=2.0).eval().to(DEV)
    unet = net.restoration_network
    xs = [torch.rand(1, 3, 264, 392, generator=gen(41)).to(DEV), torch.rand(2, 3, 136, 200, generator=gen(42)).to(DEV)]
    assert unet.two_streams
    try:
        with torch.no_grad():
            unet.two_streams = False
            refs = [unet(x) for x in xs]
            unet.two_streams = True
            for _ in range(3):                                       # back to back: no synchronisation between forwards
                outs = [unet(x) for x in xs]
            other = torch.cuda.Stream(DEV)
            other.wait_stream(torch.cuda.current_stream(DEV))
            with torch.cuda.stream(other):
                outs_other = [unet(x) for x in xs]
            torch.cuda.current_stream(DEV).wait_stream(other)
    finally:
        unet.two_streams = True
    torch.cuda.synchronize()
    for r, o, oo in zip(refs, outs, outs_other):
        assert torch.equal(r, o), f"max |diff| {float((r - o).abs().max()):.3e}"
        assert torch.equal(r, oo), f"from a non-default stream: max |diff| {float((r - oo).abs().max()):.3e}"


@pytest.mark.parametrize("planes", ["fp32", "bf16"])
def test_multi_stream_uhd_bit_equal(planes):
    """The multi-stream forward at the size of record, both plane storage types: 20 back-to-back forwards (no host
    synchronisation in between) alternating between the padded UHD frame and a 1088 x 1920 frame, every one bit-equal to the
    single-stream order.  This is the reproducer of rounds 4-5 (tools/debug_bf16_determinism.py: bf16 planes differed by 1e-2 ..
    5e-2 on EVERY multi-stream forward, fp32 planes rarely): dwconv3x3<bf16> contained `v_pk_fma_f32 v, s[..], v, v op_sel:[0,0,1]`,
    which returns zero for the routed half in lanes 48..63 while a 3x3 matrix-core convolution of a side stream shares the SIMD
    (tools/ubench_pk_coexec.hip).  VERDICT r4 item 1."""
    net = wm.WaveMamba(in_chn=3, wf=32, n_l_blocks=[1, 2, 4], n_h_blocks=[1, 1, 2], ffn_scale=2.0).eval().to(DEV)
    unet = net.restoration_network
    xs = [torch.rand(1, 3, 2176, 3840, generator=gen(1234)).to(DEV), torch.rand(1, 3, 1088, 1920, generator=gen(77)).to(DEV)]
    prev = wm.ops.set_plane_dtype(torch.bfloat16 if planes == "bf16" else torch.float32)
    try:
        with torch.no_grad():
            unet.two_streams = False
            refs = [unet(x) for x in xs]
            torch.cuda.synchronize()
            unet.two_streams = True
            outs = [unet(xs[i % 2]) for i in range(20)]              # back to back: nothing waits for anything on the host
            torch.cuda.synchronize()
    finally:
        unet.two_streams = True
        wm.ops.set_plane_dtype(prev)
    bad = [(i, float((o - refs[i % 2]).abs().max())) for i, o in enumerate(outs) if not torch.equal(o, refs[i % 2])]
    assert not bad, f"{planes} planes: {len(bad)} of 20 multi-stream forwards differ from the single-stream order: {bad[:5]}"


@pytest.mark.parametrize("two_streams", [False, True], ids=["single_stream", "multi_stream"])
def test_hip_graph_capture_after_a_multi_stream_forward(two_streams):
    """torch.cuda.graph (global capture mode) of the forward after an eager multi-stream forward has filled the prepared-weight
    caches from its SIDE streams: the capture stream finds buffers whose completion events were recorded outside the capture.
    The host waits for them through wm_event_synchronize_relaxed (a plain event.synchronize() there returned
    hipErrorStreamCaptureUnsupported and every later launch hipErrorStreamCaptureInvalidated: bench.py --graph, round 5).
    The replay is bit-equal to the eager single-stream forward."""
    net = wm.WaveMamba(in_chn=3, wf=32, n_l_blocks=[1, 2, 4], n_h_blocks=[1, 1, 2], ffn_scale=2.0).eval().to(DEV)
    unet = net.restoration_network
    x = torch.rand(1, 3, 1088, 1920, generator=gen(5)).to(DEV)
    try:
        with torch.no_grad():
            unet.two_streams = True
            unet(x)                                                  # caches filled from the side streams
            unet.two_streams = False
            ref = unet(x)
            torch.cuda.synchronize()
            unet.two_streams = two_streams
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = unet(x)
            g.replay(); g.replay()
            torch.cuda.synchronize()
    finally:
        unet.two_streams = True
    assert torch.equal(out, ref), f"max |diff| {float((out - ref).abs().max()):.3e}"


def test_prepared_weights_of_another_stream_under_capture():
    """The smallest form of the case above: a convolution weight prepared on a side stream (cache entry + event), then the
    same convolution captured on torch's capture stream."""
    g0 = gen(3)
    x = torch.randn(1, 32, 64, 96, generator=g0).to(DEV)
    w = torch.nn.Parameter((torch.randn(32, 32, 3, 3, generator=g0) / 16).to(DEV), requires_grad=False)
    side = torch.cuda.Stream(DEV)
    side.wait_stream(torch.cuda.current_stream(DEV))
    with torch.no_grad():
        with torch.cuda.stream(side):
            ref = wm.ops.conv2d(x, w)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = wm.ops.conv2d(x, w)
        g.replay()
        torch.cuda.synchronize()
    assert torch.equal(out, ref)


def test_zero_arena_skips_the_memset_only_inside_registered_ranges():
    """wm_plane_sums adds into its output: inside a registered range the library takes the buffer as zero (a buffer of ones comes
    back as sums + 1 - the memset node was skipped), outside of it and after wm_zero_arena_unregister it zeroes it itself."""
    from wave_mamba_amd.ops import _ptr, _stream
    lib = wm._lib.load()
    x = torch.randn(2, 8, 16, 32, generator=gen(1)).to(DEV)
    ref = x.double().sum((0, 2, 3)).float()
    arena = torch.ones(1024, device=DEV)
    out_in = arena[64:72]
    outside = torch.ones(8, device=DEV)
    assert lib.wm_zero_arena_register(arena.data_ptr(), 4096) == 0
    try:
        for o in (out_in, outside):
            assert lib.wm_plane_sums(_ptr(x), _ptr(o), 2, 8, 16, 32, _stream()) == 0
        torch.cuda.synchronize()
        assert_close(out_in, ref + 1.0, 1e-5, "inside a registered range: taken as zero")
        assert_close(outside, ref, 1e-5, "outside: zeroed by the library")
        assert lib.wm_zero_arena_unregister(arena.data_ptr()) == 0
        assert lib.wm_zero_arena_register(arena.data_ptr(), 64 * 4 + 16) == 0  # ends in the middle of out_in: only PARTLY inside
        arena.fill_(1.0)
        assert lib.wm_plane_sums(_ptr(x), _ptr(out_in), 2, 8, 16, 32, _stream()) == 0
        torch.cuda.synchronize()
        assert_close(out_in, ref, 1e-5, "partly inside: zeroed by the library")
    finally:
        assert lib.wm_zero_arena_unregister(arena.data_ptr()) == 0
    arena.fill_(1.0)
    assert lib.wm_plane_sums(_ptr(x), _ptr(out_in), 2, 8, 16, 32, _stream()) == 0
    torch.cuda.synchronize()
    assert_close(out_in, ref, 1e-5, "after unregister: zeroed by the library")


def test_zero_arena_clear_unregisters_and_restarts():
    """ops.zero_arena_clear(): every (device, stream) entry goes, its block is unregistered (the library zeroes buffers
    inside it itself again), the next small output comes zeroed from a fresh registered block."""
    from wave_mamba_amd import ops
    from wave_mamba_amd.ops import _ptr, _stream
    lib = wm._lib.load()
    x = torch.randn(2, 8, 16, 32, generator=gen(2)).to(DEV)
    ref = x.double().sum((0, 2, 3)).float()
    side = torch.cuda.Stream(DEV)
    with torch.cuda.stream(side):
        ops._zeros_small(8, torch.device(DEV))
    side.synchronize()
    old = ops._zeros_small(8, torch.device(DEV))
    assert len(ops._ZERO_ARENAS) >= 2
    old.fill_(1.0)
    ops.zero_arena_clear()
    assert len(ops._ZERO_ARENAS) == 0
    assert lib.wm_plane_sums(_ptr(x), _ptr(old), 2, 8, 16, 32, _stream()) == 0     # no longer inside a registered range: zeroed first
    torch.cuda.synchronize()
    assert_close(old, ref, 1e-5, "a slice of a cleared arena is an ordinary buffer again")
    new = ops._zeros_small(8, torch.device(DEV))
    assert float(new.abs().max()) == 0.0 and new.untyped_storage().data_ptr() != old.untyped_storage().data_ptr()
    assert_close(ops.plane_sums(x), ref, 1e-5, "plane_sums after the clear")


def test_zeros_small_hands_out_zeroed_distinct_slices():
    """ops._zeros_small: slices are zero, 256-byte slots, never handed out twice - across the switch to a fresh block too - and a
    gradient computed into one (plane_sums) matches float64."""
    from wave_mamba_amd import ops
    seen = set()
    n_blocks = set()
    for i in range(3000):                                                      # 3000 x 2 KB > one 4-MB block
        t = ops._zeros_small(500, DEV)
        assert t.numel() == 500 and t.data_ptr() % 256 == 0 and t.data_ptr() not in seen
        seen.add(t.data_ptr())
        n_blocks.add(t.untyped_storage().data_ptr())
        if i % 500 == 0:
            assert float(t.abs().max()) == 0.0
            t.fill_(3.0)                                                       # dirtying a slice must not reach a later one
    assert len(n_blocks) >= 2
    big = ops._zeros_small(1 << 20, DEV)                                       # beyond the arena's slice limit: plain memory
    assert big.numel() == 1 << 20
    x = torch.randn(4, 32, 24, 40, generator=gen(2)).to(DEV)
    assert_close(ops.plane_sums(x), x.double().sum((0, 2, 3)).float(), 1e-5, "plane_sums into an arena slice")
    other = torch.cuda.Stream(DEV)
    other.wait_stream(torch.cuda.current_stream(DEV))
    with torch.cuda.stream(other):                                             # another stream: its own block
        t2 = ops._zeros_small(16, DEV)
        assert t2.untyped_storage().data_ptr() not in n_blocks and float(t2.abs().max()) == 0.0
    torch.cuda.current_stream(DEV).wait_stream(other)


def test_cuda_tensors_fail_loudly_without_the_library(monkeypatch):
    """The CPU twins of dwt_init / iwt_init / selective_scan_fn (cpu_twin.py) are selected by the input's device and by nothing
    else: with the HIP library gone, a CUDA tensor raises WaveMambaHipError - it is not quietly computed somewhere else."""
    from wave_mamba_amd import _lib, cpu_twin

    def boom(*a, **k):
        raise AssertionError("the CPU twin was reached by a CUDA tensor")
    for name in ("dwt_init", "iwt_init", "iwt_init_pair", "selective_scan_fn"):
        monkeypatch.setattr(cpu_twin, name, boom)
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libwavemamba_hip.so")
    x = torch.zeros(1, 4, 8, 8, device=DEV)
    for call in (lambda: wm.ops.dwt_init(x), lambda: wm.ops.iwt_init(x), lambda: wm.ops.iwt_init_pair(x[:, :1], x[:, 1:]),
                 lambda: wm.ops.selective_scan_fn(torch.zeros(1, 4, 8, device=DEV), torch.zeros(1, 4, 8, device=DEV),
                                                  -torch.ones(4, 2, device=DEV), torch.zeros(1, 1, 2, 8, device=DEV),
                                                  torch.zeros(1, 1, 2, 8, device=DEV))):
        with pytest.raises(_lib.WaveMambaHipError):
            call()


def test_graphed_train_step_matches_the_eager_step():
    """trainer.GraphedTrainStep: optimize_parameters() captured into a HIP graph and replayed on new batches - the same losses and
    parameters as the eager steps on the same batches (atomics in the gradient kernels: equal to ~1e-6, not bit for bit)."""
    cfg = dict(in_chn=3, wf=16, n_l_blocks=[1, 1, 1], n_h_blocks=[1, 1, 1], ffn_scale=2.0)
    gg = gen(77)
    batches = [(torch.rand(2, 3, 64, 64, generator=gg).to(DEV), torch.rand(2, 3, 64, 64, generator=gg).to(DEV)) for _ in range(3)]

    def fresh():
        torch.manual_seed(0)
        net = wm.WaveMamba(**cfg).train().to(DEV)
        return net, wm.trainer.make_optimizer(net, capturable=True)
    net_e, opt_e = fresh()
    for _ in range(3):                                             # the graphed step's warm-up, eagerly
        wm.trainer.train_step(net_e, opt_e, *batches[0], as_float=False)
    want = [wm.trainer.loss_values(wm.trainer.train_step(net_e, opt_e, lq, gt, as_float=False)) for lq, gt in batches]
    net_g, opt_g = fresh()
    step = wm.trainer.GraphedTrainStep(net_g, opt_g, *batches[0])
    got = [wm.trainer.loss_values(step(lq, gt)) for lq, gt in batches]
    for a, b in zip(got, want):
        for k in a:
            assert abs(a[k] - b[k]) <= 1e-5 * abs(b[k]), f"{k}: graphed {a[k]} eager {b[k]}"
    worst = max(float((p.detach() - q.detach()).abs().max() / (q.detach().abs().max() + 1e-12))
                for p, q in zip(net_g.parameters(), net_e.parameters()))
    assert worst <= 1e-3, f"parameters after 3 + 3 steps differ by {worst:.2e}"
    with pytest.raises(RuntimeError):
        wm.trainer.GraphedTrainStep(net_e, wm.trainer.make_optimizer(net_e), *batches[0])       # not capturable


def _eager_noise():
    """What a training loop does between two replays: kernels and allocations of its own (NaN / 1e30 fills of several sizes)."""
    for n, v in ((8, float("nan")), (320, 1e30), (1024, float("nan")), (65536, 1e30), (1 << 20, float("nan"))):
        t = torch.full((n,), v, device=DEV)
        del t
    torch.cuda.synchronize()


def test_graph_replay_zeroes_its_accumulators_with_eager_launches_between_replays():
    """Round 6: accumulate-into outputs were zeroed by hipMemsetAsync; captured into a HIP graph that is a memset node whose fill
    pattern this runtime re-reads, at every launch of the graph, from memory it has recycled - any eager kernel launch between two
    replays and the node filled the gradient buffers with other kernels' arguments (tools/repro_graph_memset_node.py).  The library
    zeroes with a kernel now.  One accumulate-into operator of each family, captured alone, replayed with eager work in between."""
    lib = wm._lib.load()
    gg = gen(5)
    B, C, H, W = 2, 32, 32, 32
    x, gy = torch.randn(B, C, H, W, generator=gg).to(DEV), torch.randn(B, C, H, W, generator=gg).to(DEV)
    lnw = torch.randn(C, generator=gg).to(DEV)

    def dw_wgrad(buf):
        wm.ops.check(lib.wm_dwconv3x3_wgrad(x.data_ptr(), gy.data_ptr(), buf[:9 * C].data_ptr(), buf[9 * C:].data_ptr(), B, C, H, W,
                                            torch.cuda.current_stream().cuda_stream), "wm_dwconv3x3_wgrad")

    def ln_bwd(buf):
        gx = torch.empty_like(x)
        wm.ops.check(lib.wm_layernorm2d_bwd(x.data_ptr(), lnw.data_ptr(), gy.data_ptr(), 1e-6, gx.data_ptr(), buf[:C].data_ptr(),
                                            buf[C:2 * C].data_ptr(), B, H * W, C, torch.cuda.current_stream().cuda_stream),
                     "wm_layernorm2d_bwd")
    for name, fn, n in (("dwconv3x3_wgrad", dw_wgrad, 10 * C), ("layernorm2d_bwd", ln_bwd, 2 * C)):
        ref = torch.empty(n, device=DEV)
        fn(ref)
        torch.cuda.synchronize()
        ref_cpu = ref.cpu()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            buf = torch.empty(n, device=DEV)
            fn(buf)
        for r in range(4):
            graph.replay()
            torch.cuda.synchronize()
            got = buf.cpu()
            err = float((got - ref_cpu).abs().max() / ref_cpu.abs().max())
            assert err <= 1e-5, f"{name}: replay {r} differs from the eager call by {err:.3e} (atomics reorder sums: ~1e-7 expected)"
            _eager_noise()


@pytest.mark.parametrize("shape", [(2, 3, 64, 64), (1, 3, 37, 53), (8, 3, 512, 512), (2, 3, 128, 65, 2), (5,)])
def test_l1_mean_vs_float64(shape):
    """ops.l1_mean = nn.L1Loss() / the reduction of FFTLoss (femasr_model.py:171-179, losses.py:306-313) on HIP kernels: value
    against the float64 mean, gradients in both arguments against ATen's (sign(a - b) * g / n, exact), odd sizes and an
    unaligned view included; bit-stable enough for logging (atomics: ~1e-7 run to run)."""
    gg = gen(3)
    a = torch.rand(*shape, generator=gg).to(DEV).requires_grad_(True)
    b = torch.rand(*shape, generator=gg).to(DEV).requires_grad_(True)
    with torch.no_grad():
        if a.numel() > 4:
            a.view(-1)[1] = b.view(-1)[1]                        # an exact tie: sign(0) = 0
    out = wm.ops.l1_mean(a, b)
    assert out.shape == ()
    want = (a.detach().double() - b.detach().double()).abs().mean()
    assert abs(float(out) - float(want)) <= 2e-6 * float(want)
    (out * 3.0).backward()
    ga, gb = a.grad.clone(), b.grad.clone()
    a2, b2 = a.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    (F.l1_loss(a2, b2) * 3.0).backward()
    assert torch.equal(ga, a2.grad) and torch.equal(gb, b2.grad)
    again = [float(wm.ops.l1_mean(a, b)) for _ in range(3)]
    assert max(abs(v - float(out)) for v in again) <= 1e-6 * float(want)
    with pytest.raises(RuntimeError):
        wm.ops.l1_mean(a, b.reshape(-1)[: max(1, b.numel() - 1)])
    with pytest.raises(RuntimeError):
        wm.ops.l1_mean(a.cpu(), b.cpu())


def test_trainer_losses_equal_the_reference_composition():
    """trainer.losses on the GPU (HIP reductions, view_as_real) = the reference's composition (F.l1_loss, stacked real / imag
    parts) in value and in the gradient that reaches the prediction."""
    gg = gen(9)
    pred = torch.rand(2, 3, 96, 160, generator=gg).to(DEV).requires_grad_(True)
    gt = torch.rand(2, 3, 96, 160, generator=gg).to(DEV)
    l_pix, l_freq = wm.trainer.losses(pred, gt)
    (l_pix + l_freq).mean().backward()
    p2 = pred.detach().clone().requires_grad_(True)
    pf, tf = torch.fft.rfft2(p2), torch.fft.rfft2(gt)
    r_pix = F.l1_loss(p2, gt)
    r_freq = 0.1 * F.l1_loss(torch.stack([pf.real, pf.imag], dim=-1), torch.stack([tf.real, tf.imag], dim=-1))
    (r_pix + r_freq).mean().backward()
    assert abs(float(l_pix) - float(r_pix)) <= 2e-6 * float(r_pix) and abs(float(l_freq) - float(r_freq)) <= 2e-6 * float(r_freq)
    assert_close(pred.grad, p2.grad, 1e-5, "d loss / d prediction")


def test_graphed_train_step_with_eager_work_between_replays():
    """The whole optimize_parameters() replayed from a graph while the caller runs kernels of its own between the replays (logging,
    checks, data preparation): the same losses as the eager steps, finite gradients, every replay (round 6: NaN after the second)."""
    _graphed_with_eager_work(dict(in_chn=3, wf=16, n_l_blocks=[1, 1, 1], n_h_blocks=[1, 1, 1], ffn_scale=2.0), 64)


def test_graphed_train_step_with_eager_work_between_replays_shipped_config():
    """... and the shipped configuration at 256 x 256 (round 6: with ATen's l1_loss in the step - its reduction zeroes semaphores with
    a memset node - the weights of a replay stayed right and its REPORTED l_pix went from 0.33 to 1.10)."""
    # (Adam's first updates are lr * sign-like in every gradient element: tensors whose gradient noise straddles zero move by up to
    # 2 lr per step either way, hence the looser bar on the parameters of the 591-tensor network; the losses carry the 1e-5 bar)
    _graphed_with_eager_work(dict(in_chn=3, wf=32, n_l_blocks=[1, 2, 4], n_h_blocks=[1, 1, 2], ffn_scale=2.0), 256, wbar=5e-2)


def _graphed_with_eager_work(cfg, size, wbar=1e-3):
    gg = gen(77)
    batches = [(torch.rand(2, 3, size, size, generator=gg).to(DEV), torch.rand(2, 3, size, size, generator=gg).to(DEV)) for _ in range(4)]

    def fresh():
        torch.manual_seed(0)
        net = wm.WaveMamba(**cfg).train().to(DEV)
        return net, wm.trainer.make_optimizer(net, capturable=True)
    net_e, opt_e = fresh()
    for _ in range(3):
        wm.trainer.train_step(net_e, opt_e, *batches[0], as_float=False)
    want = [wm.trainer.loss_values(wm.trainer.train_step(net_e, opt_e, lq, gt, as_float=False)) for lq, gt in batches]
    net_g, opt_g = fresh()
    step = wm.trainer.GraphedTrainStep(net_g, opt_g, *batches[0])
    for (lq, gt), w in zip(batches, want):
        got = wm.trainer.loss_values(step(lq, gt))
        bad = [n for n, p in net_g.named_parameters() if not bool(torch.isfinite(p.grad).all())]
        assert not bad, f"non-finite gradients after a replay: {bad[:5]}"
        for k in got:
            assert abs(got[k] - w[k]) <= 1e-5 * abs(w[k]), f"{k}: graphed {got[k]} eager {w[k]}"
        _eager_noise()
    worst = max(float((p.detach() - q.detach()).abs().max() / (q.detach().abs().max() + 1e-12))
                for p, q in zip(net_g.parameters(), net_e.parameters()))
    assert worst <= wbar, f"parameters after 3 + 4 steps differ by {worst:.2e}"


def test_graphed_train_step_follows_a_learning_rate_schedule():
    """a float lr would be frozen into the captured AdamW launch.  make_optimizer(capturable=True) keeps lr as a device
    tensor; a torch scheduler stepped between replays fills it in place and the replayed update follows: graphed + scheduler ==
    eager + scheduler, and the schedule visibly changes the update (lr -> 0 leaves only nothing: weight decay scales with lr too)."""
    cfg = dict(in_chn=3, wf=8, n_l_blocks=[1, 1, 1], n_h_blocks=[1, 1, 1], ffn_scale=2.0)
    gg = gen(78)
    lq, gt = torch.rand(2, 3, 64, 64, generator=gg).to(DEV), torch.rand(2, 3, 64, 64, generator=gg).to(DEV)

    def fresh():
        torch.manual_seed(0)
        net = wm.WaveMamba(**cfg).train().to(DEV)
        opt = wm.trainer.make_optimizer(net, capturable=True)
        return net, opt, torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.0)    # lr, 0, 0, ...
    net_e, opt_e, sch_e = fresh()
    assert isinstance(opt_e.param_groups[0]["lr"], torch.Tensor) and opt_e.param_groups[0]["lr"].is_cuda
    for _ in range(3):
        wm.trainer.train_step(net_e, opt_e, lq, gt, as_float=False)
    wm.trainer.train_step(net_e, opt_e, lq, gt, as_float=False); sch_e.step()
    after_first = [p.detach().clone() for p in net_e.parameters()]
    wm.trainer.train_step(net_e, opt_e, lq, gt, as_float=False)                           # lr = 0: nothing moves
    net_g, opt_g, sch_g = fresh()
    step = wm.trainer.GraphedTrainStep(net_g, opt_g, lq, gt)
    step(lq, gt); sch_g.step()
    assert float(opt_g.param_groups[0]["lr"]) == 0.0
    mid = [p.detach().clone() for p in net_g.parameters()]
    step(lq, gt)
    torch.cuda.synchronize()
    for p, m in zip(net_g.parameters(), mid):
        assert torch.equal(p.detach(), m), "a replay at lr = 0 moved a parameter: the captured lr is not the scheduler's"
    worst = max(float((p.detach() - q).abs().max() / (q.abs().max() + 1e-12)) for p, q in zip(net_g.parameters(), after_first))
    assert worst <= 1e-3, f"graphed + scheduler differs from eager + scheduler by {worst:.2e}"
    with pytest.raises(RuntimeError):                                                     # a float lr is refused, not ignored
        bad = torch.optim.AdamW(net_e.parameters(), lr=5e-4, capturable=True, fused=True)
        wm.trainer.GraphedTrainStep(net_e, bad, lq, gt)


def _concurrency_victims(level_hw):
    """Operator groups of the shipped network on fixed inputs (tools/repro_victim_sweep.py): name -> callable."""
    H, W = level_hw
    net = wm.WaveMamba(in_chn=3, wf=32, n_l_blocks=[1, 2, 4], n_h_blocks=[1, 1, 2], ffn_scale=2.0).eval().to(DEV)
    unet = net.restoration_network
    dg, ug = unet.down_group3, unet.up_group3
    blk = dg.l_blk[0]
    ss = blk.self_attention
    gg = gen(11)
    r = lambda *shape: torch.randn(*shape, generator=gg).to(DEV)
    x32, x64, low, x96, full = r(1, 32, H, W), r(1, 64, H, W), r(1, 32, H, W), r(1, 96, H, W), r(1, 32, 2 * H, 2 * W)
    hl, lh, hh = r(1, 32, H, W), r(1, 32, H, W), r(1, 32, H, W)
    xb = x64.bfloat16()

    def lfss(dt):
        def f():
            prev = wm.ops.set_plane_dtype(dt)
            try:
                return wm.ops.lfss_block_forward(x32, (H, W), blk, tok_nchw=True, out_nchw=True)
            finally:
                wm.ops.set_plane_dtype(prev)
        return f
    hb = dg.h_blk[0]
    return {
        "lfss_block fp32 planes": lfss(torch.float32),
        "lfss_block bf16 planes": lfss(torch.bfloat16),
        "dwconv3x3+silu fp32": lambda: wm.ops.dwconv3x3(x64, ss.conv2d.weight, ss.conv2d.bias, "silu"),
        "dwconv3x3+silu bf16": lambda: wm.ops.dwconv3x3(xb, ss.conv2d.weight, ss.conv2d.bias, "silu"),
        "dwconv3x3 bf16": lambda: wm.ops.dwconv3x3(xb, ss.conv2d.weight, ss.conv2d.bias, "none"),
        "dwconv3x3+gelu fp32": lambda: wm.ops.dwconv3x3(x32, hb.ffn.project_out[0].weight, hb.ffn.project_out[0].bias, "gelu"),
        "HFEBlock": lambda: hb(x32, low),
        "SKFF": lambda: dg.h_fusion([hl, lh, hh]),
        "dwt": lambda: wm.ops.dwt_init(full),
        "dwt bf16": lambda: wm.ops.dwt_init(full.bfloat16()),
        "iwt pair": lambda: wm.ops.iwt_init_pair(x32, x96),
        "conv3x3 64->32 (cat)": lambda: wm.ops.conv2d(x32, dg.l_conv.weight, dg.l_conv.bias, low),
        "conv3x3 32->96": lambda: wm.ops.conv2d(x32, ug.h_out_conv.weight, ug.h_out_conv.bias),
        "patchify r=8": lambda: wm.ops.patchify_conv(_PATCH_IMG[0], unet.ps_down3[1].weight, unet.ps_down3[1].bias, 8),
        "layernorm2d": lambda: wm.ops.layernorm2d(x32, hb.LayerNorm.weight, hb.LayerNorm.bias, hb.LayerNorm.eps),
        "gram": lambda: wm.ops.gram(x32.flatten(2), low.flatten(2)),
    }


_PATCH_IMG = []


@pytest.mark.parametrize("aggressor", ["conv3x3", "conv3x3_first_generation"])
def test_kernels_unaffected_by_a_concurrent_conv3x3(aggressor):
    """No kernel of the inference forward may change its result because another kernel shares the compute units: every operator
    group of the shipped network (level-3 maps: the grids that leave room for a second kernel), 60 launches each on fixed inputs
    while a 3x3 matrix-core convolution (LDS-fed MFMAs - the measured trigger) runs on a second stream, each launch compared bit
    for bit with the launch that ran alone.  Nothing is shared between the streams, so any difference is the hardware hazard of
    tools/ubench_pk_coexec.hip (or a new one): round 4's library failed here in dwconv3x3<bf16> on 50-95 % of the launches
    (profiles/r05/victim_sweep_round4_library.txt)."""
    H, W = 272, 480
    if not _PATCH_IMG:
        _PATCH_IMG.append(torch.rand(1, 3, 8 * H, 8 * W, generator=gen(5)).to(DEV))
    victims = _concurrency_victims((H, W))
    xa = torch.randn(1, 64, 544, 960, generator=gen(12)).to(DEV)
    w3 = (torch.randn(64, 64, 3, 3, generator=gen(13)) / 24).to(DEV)
    side = torch.cuda.Stream(DEV)

    def bits(t):
        return t.view(torch.int16) if t.dtype == torch.bfloat16 else t

    def flat(o):
        return [o] if isinstance(o, torch.Tensor) else [t for v in o for t in flat(v)]
    wm.ops.conv2d_select(wm.ops.CONV3X3_FIRST_GEN if aggressor == "conv3x3_first_generation" else wm.ops.CONV3X3_WAVE_SPECIALISED)
    failures = {}
    try:
        with torch.no_grad():
            for name, v in victims.items():
                ref = [t.clone() for t in flat(v())]
                torch.cuda.synchronize()
                cnts, keep = [], []
                for _ in range(60):
                    with torch.cuda.stream(side):
                        keep.append(wm.ops.conv2d(xa, w3))
                        if len(keep) > 6:
                            keep.pop(0)
                    cnts.append(sum((bits(p) != bits(q)).sum() for p, q in zip(flat(v()), ref)))
                torch.cuda.synchronize()
                nbad = sum(1 for c in cnts if int(c))
                if nbad:
                    failures[name] = nbad
    finally:
        wm.ops.conv2d_select(wm.ops.CONV3X3_AUTO)
    assert not failures, f"launches (of 60) whose result changed under a concurrent {aggressor}: {failures}"


@pytest.mark.parametrize("B,Ca,Csrc,Cb,Cout,H,W", [(1, 32, 32, 32, 64, 24, 40), (2, 32, 32, 32, 64, 9, 33),
                                                  (1, 64, 0, 0, 64, 31, 17), (2, 16, 24, 8, 40, 17, 31)])
def test_conv2d_gated_vs_torch(conv3x3_impl, B, Ca, Csrc, Cb, Cout, H, W):
    """PAConv's k3(x) * sigmoid(k2(x)) (reference :694-697) in one kernel, on cat([x, gather(p, idx)])."""
    import torch.nn.functional as F
    gg = gen(B * 17 + Cout)
    x = torch.randn(B, Ca, H, W, generator=gg)
    w3 = torch.randn(Cout, Ca + Cb, 3, 3, generator=gg) / (3 * (Ca + Cb) ** 0.5)
    w1 = torch.randn(Cout, Ca + Cb, 1, 1, generator=gg) / (Ca + Cb) ** 0.5
    b1 = torch.randn(Cout, generator=gg)
    if Cb:
        p = torch.randn(B, Csrc, H, W, generator=gg)
        idx = torch.randint(0, Csrc, (B, Cb), generator=gg)
        xin = torch.cat([x, torch.gather(p, 1, idx[:, :, None, None].expand(-1, -1, H, W))], 1)
        got = wm.ops.conv2d_gated(*cu(x, w3, w1, b1, p, idx))
    else:
        xin = x
        got = wm.ops.conv2d_gated(*cu(x, w3, w1, b1))
    ref = F.conv2d(xin.double(), w3.double(), None, padding=1) * torch.sigmoid(F.conv2d(xin.double(), w1.double(), b1.double()))
    assert_close(got, ref.float(), 2e-5, f"conv2d_gated {(B, Ca, Cb, Cout, H, W)}")


@pytest.mark.parametrize("B,Ca,Csrc,Cb,Cout,H,W", [
    (1, 64, 0, 0, 64, 136, 256),       # interior tiles only (64 x 8 tiles divide the image): the predicate-free stores
    (2, 32, 32, 32, 64, 70, 150),      # ragged right / bottom tiles, concatenated + gathered operand, batch 2
    (1, 3, 0, 0, 32, 50, 130),         # one partial 16-channel chunk
    (1, 32, 0, 0, 3, 64, 192),         # a partial row tile (channel guards)
    (1, 80, 40, 24, 96, 40, 70),       # five chunks + a half, three row tiles (a 2-tile and a 1-tile launch)
    (3, 16, 8, 8, 40, 7, 5)])          # an image smaller than a tile
def test_conv3x3_kernels_bit_identical(B, Ca, Csrc, Cb, Cout, H, W):
    """The persistent wave-specialised 3x3 (conv2d_ws.hip.h) against the first-generation kernel on the same inputs, every
    fused form: same MFMA sequence per output element -> equal bit for bit (any difference is an indexing / hand-off bug,
    not rounding).  Each form is also run twice on the wave-specialised kernel (a race between producer and consumer
    waves would show as run-to-run differences)."""
    gg = gen(B * 7 + Ca + Cout + W)
    x = torch.randn(B, Ca, H, W, generator=gg)
    p = torch.randn(B, Csrc, H, W, generator=gg) if Cb else None
    idx = torch.randint(0, Csrc, (B, Cb), generator=gg) if Cb else None
    w3 = torch.randn(Cout, Ca + Cb, 3, 3, generator=gg) / (3 * (Ca + Cb) ** 0.5)
    w1 = torch.randn(Cout, Ca + Cb, 1, 1, generator=gg) / (Ca + Cb) ** 0.5
    b = torch.randn(Cout, generator=gg)
    gate = torch.randn(B, Cout, H, W, generator=gg)
    res = torch.randn(B, Cout, H, W, generator=gg)
    xd, pd, idxd, w3d, w1d, bd, gd, rd = cu(x, p, idx, w3, w1, b, gate, res)
    forms = {
        "plain": lambda: wm.ops.conv2d(xd, w3d, bd, pd, idxd),
        "no bias": lambda: wm.ops.conv2d(xd, w3d, None, pd, idxd),
        "gate": lambda: wm.ops.conv2d(xd, w3d, bd, pd, idxd, gate=gd),
        "residual": lambda: wm.ops.conv2d(xd, w3d, bd, pd, idxd, residual=rd),
        "k3 * sigmoid(k2)": lambda: wm.ops.conv2d_gated(xd, w3d, w1d, bd, pd, idxd),
    }
    try:
        for name, fn in forms.items():
            wm.ops.conv2d_select(wm.ops.CONV3X3_FIRST_GEN)
            ref = fn()
            wm.ops.conv2d_select(wm.ops.CONV3X3_WAVE_SPECIALISED)
            got, again = fn(), fn()
            assert torch.equal(got, ref), f"{name}: max |diff| {float((got - ref).abs().max()):.3e}"
            assert torch.equal(got, again), f"{name}: the wave-specialised kernel is not reproducible"
    finally:
        wm.ops.conv2d_select(wm.ops.CONV3X3_AUTO)


def test_conv3x3_auto_choice_at_uhd_level1_is_bit_identical_too():
    """At UHD level 1 the automatic choice is the wave-specialised kernel (16 tiles per compute unit, the chunk pipeline
    running across tile boundaries): same bits as the first-generation kernel on a 64 -> 64 and a gated 64 -> 64."""
    gg = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(1, 64, 1088, 1920, device=DEV, generator=gg)
    w3 = torch.randn(64, 64, 3, 3, device=DEV, generator=gg) / 24
    w1 = torch.randn(64, 64, 1, 1, device=DEV, generator=gg) / 8
    b1 = torch.randn(64, device=DEV, generator=gg)
    try:
        auto = wm.ops.conv2d(x, w3), wm.ops.conv2d_gated(x, w3, w1, b1)
        wm.ops.conv2d_select(wm.ops.CONV3X3_FIRST_GEN)
        ref = wm.ops.conv2d(x, w3), wm.ops.conv2d_gated(x, w3, w1, b1)
    finally:
        wm.ops.conv2d_select(wm.ops.CONV3X3_AUTO)
    assert torch.equal(auto[0], ref[0]) and torch.equal(auto[1], ref[1])


@pytest.mark.parametrize("shape", [(2, 40, 56, 32), (1, 7, 9, 64), (3, 130, 16), (5, 8)])
def test_layernorm_tok_forward_backward_vs_torch(shape):
    """nn.LayerNorm over the last axis of token tensors (ln_1 / ln_2 / out_norm), forward and gradients."""
    import torch.nn.functional as F
    C = shape[-1]
    gg = gen(C + len(shape))
    x = torch.randn(*shape, generator=gg) * 2.0 + 0.5
    w = torch.randn(C, generator=gg) * 0.5 + 1.0
    b = torch.randn(C, generator=gg)
    gy = torch.randn(*shape, generator=gg)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    ref = F.layer_norm(xr, (C,), wr, br, 1e-5)
    gref = torch.autograd.grad(ref, (xr, wr, br), gy.double())
    xd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    got = wm.ops.layernorm_tok(xd, wd, bd, 1e-5)
    ggot = torch.autograd.grad(got, (xd, wd, bd), gy.to(DEV))
    assert_close(got.detach(), ref.detach().float(), 1e-5, f"layernorm_tok {shape}")
    for a, r, nm in zip(ggot, gref, ("gx", "dweight", "dbias")):
        assert_close(a, r.float(), 2e-5, f"layernorm_tok {nm} {shape}")


# ------------------------------------------------------------------------------------------------
# the caller's I/O step: uint8 <-> padded fp32 tensors, bit-exact against the PyTorch spelling
# of inference_wavemamba.py:99-113 / img_util.py
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("h,w", [(70, 130), (128, 256), (130, 200), (257, 129)])
@pytest.mark.parametrize("swap", [True, False])
def test_image_pre_post_u8_bit_exact(h, w, swap):
    from wave_mamba_amd import inference
    gg = gen(h + w)
    img = torch.randint(0, 256, (h, w, 3), generator=gg, dtype=torch.uint8)
    t = img.permute(2, 0, 1).float()
    if swap:
        t = t[[2, 1, 0]]
    ref = inference.check_image_size((t / 255.0).unsqueeze(0))
    got = wm.ops.image_pre_u8(img.to(DEV), 128, swap)
    assert torch.equal(got.cpu(), ref)
    y = torch.randn(1, 3, ref.shape[2], ref.shape[3], generator=gg) * 0.7 + 0.5
    y[0, 0, 0, :4] = torch.tensor([0.5 / 255, 1.5 / 255, 2.5 / 255, 254.5 / 255])      # ties: half to even
    q = (y[:, :, :h, :w].clamp(0, 1) * 255.0).round().to(torch.uint8)[0]
    if swap:
        q = q[[2, 1, 0]]
    assert torch.equal(wm.ops.image_post_u8(y.to(DEV), h, w, swap).cpu(), q.permute(1, 2, 0).contiguous())


def test_uint8_pipeline_matches_sequential(golden):
    from wave_mamba_amd import inference
    import numpy as np
    torch.manual_seed(0)
    net = wm.WaveMamba(in_chn=3, wf=8, n_l_blocks=[1, 1, 1], n_h_blocks=[1, 1, 1], ffn_scale=2.0).eval().to(DEV)
    rng = np.random.default_rng(0)
    imgs = [rng.integers(0, 256, size=(90 + 7 * i, 150 - 5 * i, 3), dtype=np.uint8) for i in range(5)]
    pipe = inference.UInt8Pipeline(net, DEV)
    outs = list(pipe.run(imgs))
    assert len(outs) == len(imgs)
    for im, o in zip(imgs, outs):
        t = torch.from_numpy(im).to(DEV)
        with torch.no_grad():
            ref = wm.ops.image_post_u8(net.restoration_network(wm.ops.image_pre_u8(t)), im.shape[0], im.shape[1])
        assert o.shape == im.shape and np.array_equal(o, ref.cpu().numpy())
    # images large enough for the GPU to lag behind the host, shapes alternating (every slot is re-allocated and re-used
    # while the previous forward is still queued): the cross-stream orderings of upload() are what keeps this exact
    big = [rng.integers(0, 256, size=((520, 776, 3) if i % 2 else (392, 1000, 3)), dtype=np.uint8) for i in range(7)]
    refs = []
    for im in big:
        with torch.no_grad():
            refs.append(wm.ops.image_post_u8(net.restoration_network(wm.ops.image_pre_u8(torch.from_numpy(im).to(DEV))),
                                             im.shape[0], im.shape[1]).cpu().numpy())
    for rep in range(2):
        outs = list(pipe.run(big))
        for o, r in zip(outs, refs):
            assert np.array_equal(o, r)


@pytest.mark.parametrize("ks,B,Cin,Cout,H,W,bias", [(3, 2, 32, 64, 24, 40, True), (1, 1, 64, 32, 17, 33, True),
                                                    (3, 1, 3, 32, 16, 32, False), (1, 2, 12, 32, 9, 20, True)])
@pytest.mark.parametrize("mode", ["auto", "f16", "aten", "bf16x3"])
def test_conv2d_train_gradients_vs_torch_autograd(ks, B, Cin, Cout, H, W, bias, mode):
    """The three training modes of the dense convolutions against the fp64 convolution: the default (fp16 split on the matrix
    cores) and ATen's fp32 at the same bar (2e-6), the split-bf16 inference kernels at 2e-5 / 5e-5."""
    fast = mode == "bf16x3"
    gg = gen(ks * 10 + Cin)
    x = torch.randn(B, Cin, H, W, generator=gg)
    w = torch.randn(Cout, Cin, ks, ks, generator=gg) / (ks * Cin ** 0.5)
    b = torch.randn(Cout, generator=gg) if bias else None
    gy = torch.randn(B, Cout, H, W, generator=gg)
    ps = [t.double().requires_grad_(True) for t in (x, w)] + ([b.double().requires_grad_(True)] if bias else [])
    ref = F.conv2d(ps[0], ps[1], ps[2] if bias else None, padding=ks // 2)
    gref = torch.autograd.grad(ref, ps, gy.double())
    qs = [t.to(DEV).requires_grad_(True) for t in (x, w)] + ([b.to(DEV).requires_grad_(True)] if bias else [])
    prev = wm.ops.set_train_conv_mode(mode)
    try:
        got = wm.ops.conv2d_train(qs[0], qs[1], qs[2] if bias else None)
        ggot = torch.autograd.grad(got, qs, gy.to(DEV))
    finally:
        wm.ops.set_train_conv_mode(prev)
    assert_close(got.detach(), ref.detach().float(), 2e-5 if fast else 2e-6, "conv2d_train forward")
    hip_gw = wm.ops.conv2d_wgrad_supported(qs[0], qs[1])     # W % 32 == 0: the weight gradient is the split-bf16 HIP kernel's
    for a, r, nm in zip(ggot, gref, ("gx", "gw", "gb")):
        bar = 5e-5 if fast else (2e-5 if (nm == "gw" and hip_gw) else 2e-6)
        assert_close(a, r.float(), bar, f"conv2d_train {nm} ks={ks}")


@pytest.mark.parametrize("ks,B,Cin,Cout,H,W,bias", [
    (3, 8, 64, 64, 256, 256, False),     # BASELINE config 3, level 1: the wave-specialised kernel, two row tiles
    (3, 8, 64, 32, 256, 256, True),      # ... one row tile
    (3, 2, 32, 96, 128, 128, True),      # h_out_conv: three row tiles (2 + 1)
    (3, 8, 64, 64, 64, 64, False),       # level 3: the first-generation kernel
    (3, 1, 3, 32, 70, 50, True), (3, 2, 32, 3, 33, 65, True),
    (1, 8, 32, 96, 256, 256, True), (1, 2, 64, 32, 17, 33, True), (1, 1, 96, 64, 40, 56, False), (1, 3, 12, 20, 9, 20, True),
])
@pytest.mark.parametrize("scale", [1.0, 3e-7, 2e4])
def test_conv2d_f16_vs_float64(ks, B, Cin, Cout, H, W, bias, scale):
    """The training form of the dense convolutions (fp16 split, per-tensor power-of-two scales) against the float64 convolution,
    next to ATen's fp32 on the same inputs: activations of magnitude `scale` (3e-7: a gradient map - far below fp16's range
    without the scale) with a heavy-tailed distribution (a few entries 1e3 x the typical one), weights of magnitude 1 / sqrt(K)."""
    gg = gen(ks + Cin + Cout + H)
    x = torch.randn(B, Cin, H, W, generator=gg)
    x = x * torch.where(torch.rand(x.shape, generator=gg) < 1e-3, 1e3, 1.0) * scale
    w = torch.randn(Cout, Cin, ks, ks, generator=gg) / (ks * Cin ** 0.5)
    b = (torch.randn(Cout, generator=gg) * scale) if bias else None
    xd, wd = x.to(DEV), w.to(DEV)
    ref = F.conv2d(xd.double(), wd.double(), None if b is None else b.to(DEV).double(), padding=ks // 2)
    got = wm.ops.conv2d_f16(xd, wd, None if b is None else b.to(DEV))
    aten = F.conv2d(xd, wd, None if b is None else b.to(DEV), padding=ks // 2)
    e_got, e_aten = rel_err(got, ref.float()), rel_err(aten, ref.float())
    print(f"conv2d_f16 ks={ks} {B}x{Cin}->{Cout} {H}x{W} scale {scale:g}: rel l2 / max {e_got[0]:.2e} / {e_got[1]:.2e} "
          f"(ATen fp32: {e_aten[0]:.2e} / {e_aten[1]:.2e})")
    assert_close(got, ref.float(), 1e-6, "conv2d_f16")
    assert torch.equal(wm.ops.conv2d_f16(xd, wd, None if b is None else b.to(DEV)), got), "conv2d_f16: not reproducible run to run"


@pytest.mark.parametrize("B,Cout,H,W,res", [(1, 96, 40, 56, False), (2, 32, 33, 65, True), (1, 64, 272, 480, False), (1, 32, 5, 7, True),
                                            (1, 96, 1088, 1920, False)])
def test_conv2d_ln_bit_identical_to_two_launches(B, Cout, H, W, res):
    """conv1x1(LayerNorm2d(x)) in one kernel (the normalisation in the 1x1 kernel's staging registers) against
    wm_layernorm2d_fwd + wm_conv2d_fwd: the same arithmetic on the same values - equal bit for bit; and against float64."""
    gg = gen(Cout + H)
    x = (torch.randn(B, 32, H, W, generator=gg) * 1.7 + 0.3).to(DEV)
    lw, lb = (torch.randn(32, generator=gg) * 0.2 + 1.0).to(DEV), (torch.randn(32, generator=gg) * 0.1).to(DEV)
    w, b = (torch.randn(Cout, 32, 1, 1, generator=gg) / 32 ** 0.5).to(DEV), torch.randn(Cout, generator=gg).to(DEV)
    r = torch.randn(B, Cout, H, W, generator=gg).to(DEV) if res else None
    one = wm.ops.conv2d_ln(x, lw, lb, 1e-6, w, b, r)
    two = wm.ops.conv2d(wm.ops.layernorm2d(x, lw, lb, 1e-6), w, b, residual=r)
    assert torch.equal(one, two), f"max abs difference {float((one - two).abs().max()):.3e}"
    if H * W <= 200000:
        xd = x.double()
        mu = xd.mean(1, keepdim=True)
        ln = (xd - mu) / ((xd - mu).pow(2).mean(1, keepdim=True) + 1e-6).sqrt() * lw.double().view(1, -1, 1, 1) + lb.double().view(1, -1, 1, 1)
        ref = F.conv2d(ln, w.double(), b.double()) + (0 if r is None else r.double())
        assert_close(one, ref.float(), 2e-5, "conv2d_ln vs float64")


@pytest.mark.parametrize("ks,Cin,Cout", [(3, 64, 64), (1, 32, 96), (3, 3, 32), (3, 32, 3), (1, 64, 32), (3, 20, 40)])
def test_conv2d_f16_input_gradient_from_the_forward_weight(ks, Cin, Cout):
    """conv2d_f16(gy, weight, dgrad=True): the fragments of weight.transpose(0, 1).flip(2, 3) read from the forward weight
    (wm_conv2d_f16_steps, dgrad = 1) - bit-identical to preparing the materialised tensor, and the float64 input gradient of F.conv2d."""
    gg = gen(ks * 100 + Cin + Cout)
    w = (torch.randn(Cout, Cin, ks, ks, generator=gg) / (Cin * ks * ks) ** 0.5).to(DEV)
    gy = torch.randn(2, Cout, 24, 40, generator=gg).to(DEV)
    got = wm.ops.conv2d_f16(gy, w, dgrad=True)
    want = wm.ops.conv2d_f16(gy, w.transpose(0, 1).flip(2, 3).contiguous())
    assert torch.equal(got, want), f"max |diff| {float((got - want).abs().max()):.3e}"
    x = torch.zeros(2, Cin, 24, 40, dtype=torch.float64, device=DEV, requires_grad=True)
    ref, = torch.autograd.grad(F.conv2d(x, w.double(), None, padding=ks // 2), x, gy.double())
    assert_close(got, ref.float(), 1e-6, "input gradient vs float64")


def test_dwconv3x3_flipped_taps():
    """wm_dwconv3x3_fwd(act + 4): the taps rotated by 180 degrees without a flipped copy of the weight - bit-identical to passing
    weight.flip(2, 3), every activation, fp32 and bf16 planes, wide and narrow maps."""
    gg = gen(21)
    for shape in ((2, 16, 20, 36), (1, 8, 33, 260), (3, 5, 64, 64)):
        x = torch.randn(*shape, generator=gg).to(DEV)
        w = torch.randn(shape[1], 1, 3, 3, generator=gg).to(DEV)
        b = torch.randn(shape[1], generator=gg).to(DEV)
        for act in ("none", "silu", "gelu"):
            assert torch.equal(wm.ops.dwconv3x3(x, w, b, act, flip=True), wm.ops.dwconv3x3(x, w.flip(2, 3).contiguous(), b, act))
        if shape[3] % 4 == 0:
            assert torch.equal(wm.ops.dwconv3x3(x.bfloat16(), w, None, "none", flip=True),
                               wm.ops.dwconv3x3(x.bfloat16(), w.flip(2, 3).contiguous(), None, "none"))


def test_conv2d_f16_degenerate_inputs():
    """all-zero input (scale falls back to 1), a single non-zero element, a constant map: exact / fp32-class results"""
    w = torch.randn(32, 16, 3, 3, generator=gen(1)).to(DEV)
    z = torch.zeros(1, 16, 8, 32, device=DEV)
    assert float(wm.ops.conv2d_f16(z, w).abs().max()) == 0.0
    one = z.clone(); one[0, 5, 3, 7] = 1.0e-20
    ref = F.conv2d(one.double(), w.double(), padding=1)
    assert_close(wm.ops.conv2d_f16(one, w), ref.float(), 1e-6, "single tiny element")
    c = torch.full((2, 16, 8, 32), 7.25, device=DEV)
    assert_close(wm.ops.conv2d_f16(c, w), F.conv2d(c.double(), w.double(), padding=1).float(), 1e-6, "constant map")


@pytest.mark.parametrize("ks,B,Cin,Cout,H,W", [
    (3, 2, 64, 64, 8, 32),       # the HFE / plumbing 3x3: four output tiles x nine taps
    (3, 1, 32, 96, 5, 64),       # h_out_conv: 96 output channels = two passes (64 + 32)
    (3, 2, 3, 32, 6, 32),        # conv_01: a 3-channel input (one ragged input tile)
    (3, 2, 64, 32, 4, 32),
    (3, 1, 32, 3, 4, 32),        # last: 3 output channels (one ragged output tile)
    (3, 1, 16, 16, 1, 32),       # a single row: both vertical taps fall outside everywhere
    (1, 2, 32, 64, 4, 64),       # in_proj
    (1, 1, 192, 32, 3, 32),      # twelve input tiles
    (1, 1, 4, 32, 2, 32),
    (1, 3, 12, 32, 5, 96),
    (1, 2, 32, 96, 3, 32),       # ffn-like 1x1: six output tiles
    (3, 8, 64, 64, 64, 64),      # BASELINE config 3, level 3: many units per wave, several blocks
])
def test_conv2d_wgrad_vs_float64(ks, B, Cin, Cout, H, W):
    """wm_conv2d_wgrad (bf16 matrix cores, split operands) against the float64 weight gradient of F.conv2d: every tap,
    ragged channel tiles, image borders in both directions; written (not accumulated) output."""
    gg = gen(ks * 1000 + Cin * 7 + Cout + H)
    x = torch.randn(B, Cin, H, W, generator=gg)
    gy = torch.randn(B, Cout, H, W, generator=gg)
    w = torch.zeros(Cout, Cin, ks, ks, dtype=torch.float64, requires_grad=True)
    ref, = torch.autograd.grad(F.conv2d(x.double(), w, None, padding=ks // 2), w, gy.double())
    if True:
        assert wm.ops.conv2d_wgrad_supported(x.to(DEV), w)
        got = wm.ops.conv2d_wgrad(gy.to(DEV), x.to(DEV), ks)
        assert_close(got, ref.float(), 1e-5, f"conv2d_wgrad ks={ks} {Cin}->{Cout} {H}x{W}")
        again = wm.ops.conv2d_wgrad(gy.to(DEV), x.to(DEV), ks)
        assert torch.equal(again, got), "conv2d_wgrad: not bit-reproducible run to run"
        # the bias gradient from the same pass over gy (round 5): dW unchanged bit for bit, db against float64
        got_w, got_b = wm.ops.conv2d_wgrad(gy.to(DEV), x.to(DEV), ks, with_bias=True)
        assert torch.equal(got_w, got), "conv2d_wgrad: dW changes when the bias gradient rides along"
        assert_close(got_b, gy.double().sum((0, 2, 3)).float(), 1e-5, f"conv2d_wgrad bias gradient ks={ks} {Cin}->{Cout} {H}x{W}")
        assert torch.equal(wm.ops.conv2d_wgrad(gy.to(DEV), x.to(DEV), ks, with_bias=True)[1], got_b), "bias gradient: not bit-reproducible"
        # the same through autograd (conv2d_train's default mode), input and bias gradients from ATen / the plane-sum kernel
        xs = x.to(DEV).requires_grad_(True); ws = (torch.randn(Cout, Cin, ks, ks, generator=gg) * 0.1).to(DEV).requires_grad_(True)
        y = wm.ops.conv2d_train(xs, ws, None)
        gx, gw = torch.autograd.grad(y, (xs, ws), gy.to(DEV))
        assert_close(gw, ref.float(), 1e-5, "conv2d_train gw")


@pytest.mark.parametrize("T,O,I", [(5000, 128, 32), (777, 32, 64), (64, 16, 16), (3, 64, 16), (100003, 32, 32)])
def test_linear_wgrad_vs_torch(T, O, I):
    """Weight gradient of SS2D.in_proj / out_proj (nn.Linear without bias) on the MFMA reduction kernel."""
    gg = gen(T % 97 + O)
    x = torch.randn(T, I, generator=gg)
    w = torch.randn(O, I, generator=gg) / I ** 0.5
    gy = torch.randn(T, O, generator=gg)
    xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    y = wm.ops.linear_nobias(xd, wd)
    gx, gw = torch.autograd.grad(y, (xd, wd), gy.to(DEV))
    assert_close(y.detach(), (x.double() @ w.double().t()).float(), 1e-5, "linear fwd")
    assert_close(gx, (gy.double() @ w.double()).float(), 1e-5, "linear gx")
    assert_close(gw, (gy.double().t() @ x.double()).float(), 2e-5, "linear gw")


def test_conv2d_uhd_level1_crop_and_linearity():
    """BASELINE config 2 size (UHD level 1, 64 -> 64, 3x3 on cat([x, gather(p, idx)])): an interior + border crop against
    the fp64 convolution of the same crop, and linearity in the input (size-independent property)."""
    import torch.nn.functional as F
    H, W = 1088, 1920
    gg = torch.Generator(device=DEV).manual_seed(21)
    x = torch.randn(1, 32, H, W, device=DEV, generator=gg)
    p = torch.randn(1, 32, H, W, device=DEV, generator=gg)
    idx = torch.randint(0, 32, (1, 32), device=DEV, generator=gg)
    w = torch.randn(64, 64, 3, 3, device=DEV, generator=gg) / 24
    y = wm.ops.conv2d(x, w, None, p, idx)
    xin = torch.cat([x, torch.gather(p, 1, idx[:, :, None, None].expand(-1, -1, H, W))], 1)
    for (h0, w0) in ((0, 0), (H - 66, W - 130), (500, 1000)):
        hs, ws = slice(max(h0 - 1, 0), min(h0 + 65, H)), slice(max(w0 - 1, 0), min(w0 + 129, W))
        ref = F.conv2d(xin[:, :, hs, ws].double(), w.double(), padding=1)
        oh, ow = h0 - hs.start, w0 - ws.start
        a = y[:, :, h0:h0 + 64, w0:w0 + 128]
        b = ref[:, :, oh:oh + 64, ow:ow + 128]
        # rows / columns at the crop border see zero padding instead of the neighbouring pixels: compare the interior,
        # and the true image border where the padding is real
        ih = slice(0 if h0 == 0 else 1, 64 if h0 + 64 >= H else 63)
        iw = slice(0 if w0 == 0 else 1, 128 if w0 + 128 >= W else 127)
        assert_close(a[:, :, ih, iw], b[:, :, ih, iw].float().cpu(), 2e-5, f"conv2d UHD crop at {(h0, w0)}")
    y2 = wm.ops.conv2d(2.0 * x, w, None, 2.0 * p, idx)
    assert_close(y2, (2.0 * y).cpu(), 1e-6, "conv2d linearity at UHD level 1")


def test_forward_is_bit_reproducible():
    """No atomics on the inference path (Gram partials and SKFF plane sums are added in block order): two forwards of
    the same input agree bit for bit, like the reference's deterministic ATen kernels."""
    torch.manual_seed(3)
    x = torch.randn(2, 32, 272 * 480, device=DEV)
    y = torch.randn(2, 32, 272 * 480, device=DEV)
    a, b = wm.ops.gram(x, y), wm.ops.gram(x, y)
    assert all(torch.equal(p, q) for p, q in zip(a, b))
    net = wm.WaveMamba(in_chn=3, wf=32, n_l_blocks=[1, 1, 1], n_h_blocks=[1, 1, 1], ffn_scale=2.0).eval().to(DEV)
    img = torch.rand(1, 3, 256, 384, device=DEV)
    with torch.no_grad():
        o1 = net.restoration_network(img)
        o2 = net.restoration_network(img)
    assert torch.equal(o1, o2)


# ------------------------------------------------------------------------------------------------
# optimizer step and checkpoints on the GPU (femasr_model.py:157-185, base_model.py:214-261)
# ------------------------------------------------------------------------------------------------
def test_optimizer_step_arithmetic_on_gpu():
    """The fused multi-tensor AdamW the trainer uses on a GPU, fed the REFERENCE's gradients, lands on the reference's
    post-step parameters (<= 1e-6 relative per tensor)."""
    from test_checkpoint import check_optimizer_arithmetic
    print(f"worst parameter deviation after optimizer_g.step(): {check_optimizer_arithmetic(DEV):.2e}")


def test_two_training_steps_on_gpu_vs_reference():
    """trainer.train_step twice from the reference's weights on the HIP training path against the reference's two
    optimize_parameters() (tests/golden/train_opt_wf8.npz).  What is well-posed to compare: Adam's first update is
    lr * g / (|g| + eps) - a sign function of every gradient element - so a parameter element whose gradient is below the
    fp32 noise of ANY implementation moves by +-lr either way; the per-tensor bars are therefore on the UPDATE
    (p1 - p0: rel-l2 <= 5e-2, i.e. < 0.07 % of the elements with a flipped sign) and on the parameters (<= 2e-5 of
    their norm), and the losses of the second step's forward - a smooth function of all of it - within 1e-5 relative."""
    import numpy as np, os
    from conftest import GOLDEN
    from test_checkpoint import _net_with_golden_grads
    net, g, o = _net_with_golden_grads(DEV)
    for p in net.parameters():
        p.grad = None
    opt = wm.trainer.make_optimizer(net)
    lq, gt = torch.from_numpy(g["lq"]).to(DEV), torch.from_numpy(g["gt"]).to(DEV)
    l1 = wm.trainer.train_step(net, opt, lq, gt)
    assert abs(l1["l_pix"] - g["losses"][0]) < 1e-6 and abs(l1["l_freq"] - g["losses"][1]) < 1e-5
    worst_u, worst_p = 0.0, 0.0
    for k, p in net.named_parameters():
        p0, p1 = torch.from_numpy(g["w." + k]).double(), torch.from_numpy(o["p1." + k]).double()
        got = p.detach().cpu().double()
        worst_u = max(worst_u, float(((got - p0) - (p1 - p0)).norm() / (p1 - p0).norm().clamp_min(1e-30)))
        worst_p = max(worst_p, float((got - p1).norm() / p1.norm().clamp_min(1e-30)))
    print(f"after step 1: worst update deviation {worst_u:.2e}, worst parameter deviation {worst_p:.2e}")
    assert worst_u <= 5e-2 and worst_p <= 2e-5
    l2 = wm.trainer.train_step(net, opt, lq, gt)
    assert abs(l2["l_pix"] - o["losses2"][0]) <= 1e-5 * o["losses2"][0], (l2, o["losses2"])
    assert abs(l2["l_freq"] - o["losses2"][1]) <= 1e-5 * o["losses2"][1], (l2, o["losses2"])


def test_checkpoint_round_trip_on_gpu(tmp_path):
    """save_network from a GPU model -> reference-format file (CPU tensors, 'params' key) -> load_network into a fresh
    GPU model -> the HIP forward is BIT-equal (prepared weight copies of the convolutions are rebuilt on load); training
    state (optimizer moments) resumes bit-equal too."""
    torch.manual_seed(0)
    cfg = dict(in_chn=3, wf=32, n_l_blocks=[1, 2, 4], n_h_blocks=[1, 1, 2], ffn_scale=2.0)
    a = wm.WaveMamba(**cfg).eval().to(DEV)
    x = torch.rand(1, 3, 128, 192, generator=gen(5)).to(DEV)
    with torch.no_grad():
        ya = a(x)
    path = wm.trainer.save_network(a, str(tmp_path / "net_g_latest.pth"), current_iter=7, epoch=1)
    blob = torch.load(path, map_location="cpu", weights_only=True)
    assert set(blob) == {"params", "iter", "epoch"} and all(v.device.type == "cpu" for v in blob["params"].values())
    torch.manual_seed(1)
    b = wm.WaveMamba(**cfg).eval().to(DEV)
    with torch.no_grad():
        assert not torch.equal(b(x), ya)                     # different weights (and its conv fragments are now cached)
    assert wm.trainer.load_network(b, path) == ([], [], [])
    with torch.no_grad():
        assert torch.equal(b(x), ya)
    # training state
    a.train(); b.train()
    oa, ob = wm.trainer.make_optimizer(a), wm.trainer.make_optimizer(b)
    lq, gt = torch.rand(1, 3, 64, 64, generator=gen(6)).to(DEV), torch.rand(1, 3, 64, 64, generator=gen(7)).to(DEV)
    wm.trainer.train_step(a, oa, lq, gt)
    wm.trainer.save_network(a, str(tmp_path / "net_g_1.pth"), 1)
    sp = wm.trainer.save_training_state(str(tmp_path / "1.state"), 0, 1, [oa])
    wm.trainer.load_network(b, str(tmp_path / "net_g_1.pth"))
    assert wm.trainer.resume_training(sp, [ob]) == (0, 1)
    for (ka, va), (kb, vb) in zip(oa.state_dict()["state"].items(), ob.state_dict()["state"].items()):
        assert all(torch.equal(va[n].cpu(), vb[n].cpu()) for n in ("exp_avg", "exp_avg_sq")), ka


def test_training_step_shipped_config_256_vs_reference(golden):
    """One reference training step (femasr_model.py:157-185) of the SHIPPED config on 2 x 3 x 256 x 256 - level-1 maps of
    128 x 128, L = 16,384: 64 backward blocks per direction, two-level carries, several weight-gradient partials per
    convolution - against the REFERENCE's own autograd (tests/golden/train_grads_shipped256.npz, make_golden_grads_256.py):
    losses, prediction, and EVERY parameter gradient (whole tensor up to 1,024 elements, else 1,024 strided elements + whole-
    tensor sums) judged against the float64 evaluation of the reference's code like every gradient test of this file."""
    import bench
    g = golden("train_grads_shipped256")
    torch.manual_seed(0)
    net = wm.WaveMamba(**bench.SHIPPED).train().to(DEV)
    lq = torch.rand(2, 3, 256, 256, generator=gen(1234)).to(DEV)
    gt = torch.rand(2, 3, 256, 256, generator=gen(4321)).to(DEV)
    pred = net(lq)
    l_pix, l_fft = wm.trainer.losses(pred, gt)
    (l_pix + l_fft).backward()
    assert abs(float(l_pix.detach()) - float(g["losses"][0])) < 1e-5 and abs(float(l_fft.detach()) - float(g["losses"][1])) < 1e-5
    assert_vs_truth(pred.detach().flatten()[::97], g["pred_sample"], g["pred_sample_f64"], "prediction")

    def sample_index(numel, sample=1024):          # make_golden_grads_256.sample_index
        if numel <= sample:
            return torch.arange(numel)
        return (torch.arange(sample, dtype=torch.int64) * numel) // sample

    bad, worst, n = [], (0.0, 0.0, None), 0
    for k, p in net.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
        flat = p.grad.detach().flatten().double().cpu()
        idx = sample_index(flat.numel())
        truth, ref = g["t." + k].double(), g["g." + k].double()
        e_got = max(rel_err(flat[idx], truth))
        e_ref = max(max(rel_err(ref, truth)), float(g["e." + k][0]), float(g["e." + k][1]))
        # whole-tensor sums against the float64 ones (catches an error outside the sampled elements)
        f = g["f." + k]
        e_sum = abs(float(flat.sum()) - float(f[3])) / max(float(f[4]), 1e-300)
        e_l2 = abs(float(flat.norm()) - float(f[5])) / max(float(f[5]), 1e-300)
        r_sum = abs(float(f[0]) - float(f[3])) / max(float(f[4]), 1e-300)
        e_got, e_ref = max(e_got, e_sum, e_l2), max(e_ref, r_sum)
        worst = max(worst, (e_got, e_ref, k))
        n += 1
        if e_got > truth_bar(e_ref):
            bad.append((e_got, e_ref, k))
    print("shipped config 256 x 256, %d tensors: worst gradient error vs float64 truth %.3e (reference fp32 %.3e) %s" % ((n,) + worst))
    assert not bad, "gradients off the float64 truth (build, reference fp32, name): " + \
        "; ".join("%.3e %.3e %s" % b for b in sorted(bad, reverse=True)[:8])


@pytest.mark.parametrize("ks,B,Cin,Cout,H,W", [
    (3, 8, 64, 64, 256, 256),    # BASELINE config 3, level 1: the HFE / plumbing 3x3 at the size the bench times it at
    (3, 8, 64, 32, 256, 256),
    (3, 8, 32, 96, 256, 256),    # h_out_conv
    (3, 8, 3, 32, 512, 512),     # conv_01 at full resolution
    (3, 8, 32, 3, 512, 512),     # last
    (3, 8, 64, 64, 128, 128),    # level 2
    (1, 8, 64, 64, 256, 256),
    (1, 8, 32, 64, 128, 128),
])
def test_conv2d_wgrad_at_training_sizes_vs_float64(ks, B, Cin, Cout, H, W):
    """wm_conv2d_wgrad at BASELINE config 3's sizes (batch 8, 512 x 512 crops: many units per wave, every block of the
    grid, the finish kernel over hundreds of partials) against the FLOAT64 weight gradient, formed on the GPU tap by tap
    as dW[o, i, ky, kx] = sum_{b, y, x} gy[b, o, y, x] x[b, i, y + ky - p, x + kx - p] with a float64 matrix product;
    ATen's fp32 weight gradient (what the kernel replaced) is the fp32 reference of the criterion."""
    gg = torch.Generator(device=DEV).manual_seed(ks * 1000 + Cin * 7 + Cout + H)
    x = torch.randn(B, Cin, H, W, device=DEV, generator=gg)
    gy = torch.randn(B, Cout, H, W, device=DEV, generator=gg)
    pad = ks // 2
    xp = F.pad(x.double(), (pad, pad, pad, pad))
    gy2 = gy.double().permute(1, 0, 2, 3).reshape(Cout, -1)
    truth = torch.empty(Cout, Cin, ks, ks, dtype=torch.float64, device=DEV)
    for ky in range(ks):
        for kx in range(ks):
            xs = xp[:, :, ky:ky + H, kx:kx + W].permute(1, 0, 2, 3).reshape(Cin, -1)
            truth[:, :, ky, kx] = gy2 @ xs.t()
    w = torch.zeros(Cout, Cin, ks, ks, device=DEV)
    assert wm.ops.conv2d_wgrad_supported(x, w)
    got = wm.ops.conv2d_wgrad(gy, x, ks)
    aten = torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [pad, pad], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    e_got, e_ref = max(rel_err(got, truth)), max(rel_err(aten, truth))
    print(f"conv2d_wgrad ks={ks} {B}x{Cin}->{Cout} {H}x{W}: {e_got:.2e} of the float64 gradient (ATen fp32: {e_ref:.2e})")
    assert e_got <= 2e-5, f"conv2d_wgrad ks={ks} {Cin}->{Cout} {H}x{W}: {e_got:.3e} vs float64"
    assert torch.equal(wm.ops.conv2d_wgrad(gy, x, ks), got), "conv2d_wgrad: not bit-reproducible run to run"
    got_w, got_b = wm.ops.conv2d_wgrad(gy, x, ks, with_bias=True)
    tb = gy.double().sum((0, 2, 3))
    # (sums of ~5e5 N(0, 1) values: |db| ~ 700 with cancellation, so the error is taken against the sum of magnitudes' scale too)
    e_b = float((got_b.double() - tb).abs().max() / gy.double().abs().sum((0, 2, 3)).max())
    print(f"   bias gradient: max |err| / sum |gy| = {e_b:.2e}")
    assert torch.equal(got_w, got) and e_b <= 1e-6, f"bias gradient at training size: {e_b:.3e}"


def test_whole_model_gradients_hip_wgrad_vs_aten_wgrad():
    """(was tools/wgrad_vs_aten.py) One training step of the shipped config on 4 x 3 x 256 x 256 with the HIP convolution
    weight gradient against the same step with ATen's: every parameter gradient within 2e-5 (they differ only in the
    weight-gradient kernels: ~4e-6 per product from the split-bf16 operands)."""
    import bench
    lq = torch.rand(4, 3, 256, 256, generator=gen(11)).to(DEV)
    gt = torch.rand(4, 3, 256, 256, generator=gen(12)).to(DEV)
    grads = []
    try:
        for hip in (True, False):
            wm.ops.set_train_conv_wgrad_hip(hip)
            torch.manual_seed(0)
            net = wm.WaveMamba(**bench.SHIPPED).train().to(DEV)
            l_pix, l_fft = wm.trainer.losses(net(lq), gt)
            (l_pix + l_fft).backward()
            grads.append({k: p.grad.detach().clone() for k, p in net.named_parameters()})
    finally:
        wm.ops.set_train_conv_wgrad_hip(True)
    worst = max((max(rel_err(grads[0][k], grads[1][k])), k) for k in grads[0])
    print("HIP vs ATen convolution weight gradient, whole model: worst tensor %.3e (%s)" % worst)
    assert worst[0] <= 2e-5, worst


def test_training_step_at_config3_size():
    """BASELINE config 3 per GPU: ONE optimize_parameters() of the shipped config on a synthetic batch of 8 x 3 x 512 x 512
    pairs on the HIP training path: the loss equals the same network's on the host with the CPU oracle as hot-path backend
    to 1e-6 relative, every parameter receives a finite gradient, the optimizer moves every parameter."""
    import bench
    torch.manual_seed(0)
    net = wm.WaveMamba(**bench.SHIPPED).train().to(DEV)
    g = gen(bench.image_seed(0))
    lq, gt = torch.rand(8, 3, 512, 512, generator=g), torch.rand(8, 3, 512, 512, generator=g)
    cores = oracle.usable_cpus(cap=1 << 20)
    torch.set_num_threads(cores); oracle.set_num_threads(cores)
    torch.manual_seed(0)
    net_cpu = wm.WaveMamba(**bench.SHIPPED).train()
    with oracle_backend.ops_backend(oracle), torch.no_grad():
        c_pix, c_fft = wm.trainer.losses(net_cpu(lq), gt)
    before = [p.detach().clone() for p in net.parameters()]
    opt = wm.trainer.make_optimizer(net)
    opt.zero_grad(set_to_none=True)
    l_pix, l_fft = wm.trainer.losses(net(lq.to(DEV)), gt.to(DEV))
    (l_pix + l_fft).backward()
    print(f"config-3 step: l_pix {float(l_pix):.7f} (CPU oracle {float(c_pix):.7f}), l_fft {float(l_fft):.7f} ({float(c_fft):.7f})")
    assert abs(float(l_pix) - float(c_pix)) <= 1e-6 * float(c_pix)
    assert abs(float(l_fft) - float(c_fft)) <= 1e-6 * float(c_fft)
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())
    opt.step()
    moved = sum(int(not torch.equal(a, b.detach())) for a, b in zip(before, net.parameters()))
    assert moved == len(before)


# ------------------------------------------------------------------------------------------------
# gates and scaled skips of the LFSSBlock training path (wm_gate_*, wm_scale_add_*): forward + backward vs fp64 autograd
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(2, 64, 16, 24), (1, 16, 5, 7), (3, 8, 9, 9)])
@pytest.mark.parametrize("act", ["silu", "gelu", "sigmoid"])
def test_gate_and_glu_gate_vs_fp64_autograd(shape, act):
    B, C, H, W = shape
    gg = gen(C * H + W)
    a, b, g = (torch.randn(B, C, H, W, generator=gg) * 2 for _ in range(3))
    f = {"silu": F.silu, "gelu": F.gelu, "sigmoid": torch.sigmoid}[act]      # sigmoid: PAConv's k3(x) * sigmoid(k2(x)) (:697-699)
    a64, b64 = a.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = f(a64) * b64
    ra, rb = torch.autograd.grad(ref, (a64, b64), g.double())
    ad, bd = a.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    out = wm.ops.gate_act(ad, bd, act)
    ga, gb = torch.autograd.grad(out, (ad, bd), g.to(DEV))
    assert_close(out.detach(), ref.detach().float(), 2e-6, f"{act} gate")
    assert_close(ga, ra.float(), 2e-6, f"{act} gate d/da"); assert_close(gb, rb.float(), 2e-6, f"{act} gate d/db")
    # chunked form: one (B, 2C, H, W) tensor in, one gradient tensor out; operands may also be channel-chunk views
    t = torch.cat([a, b], 1).to(DEV).requires_grad_(True)
    out2 = wm.ops.glu_gate(t, act)
    (gt,) = torch.autograd.grad(out2, t, g.to(DEV))
    assert torch.equal(out2, out) and torch.equal(gt[:, :C], ga) and torch.equal(gt[:, C:], gb)
    v1, v2 = t.detach().chunk(2, dim=1)
    assert torch.equal(wm.ops.gate_act(v1, v2, act), out)


@pytest.mark.parametrize("shape", [(2, 32, 16, 24), (1, 8, 5, 7), (8, 32, 64, 64)])
def test_scale_add_vs_fp64_autograd(shape):
    B, C, H, W = shape
    gg = gen(B * 100 + W)
    x, o, g = (torch.randn(B, C, H, W, generator=gg) for _ in range(3))
    s = torch.randn(C, generator=gg) * 0.3 + 1
    x64, s64, o64 = (t.double().requires_grad_(True) for t in (x, s, o))
    ref = x64 * s64.view(1, -1, 1, 1) + o64
    rx, rs, ro = torch.autograd.grad(ref, (x64, s64, o64), g.double())
    xd, sd, od = (t.to(DEV).requires_grad_(True) for t in (x, s, o))
    out = wm.ops.scale_add(xd, sd, od)
    gx, gs, go = torch.autograd.grad(out, (xd, sd, od), g.to(DEV))
    assert_close(out.detach(), ref.detach().float(), 1e-6, "scale_add")
    assert_close(gx, rx.float(), 1e-6, "scale_add d/dx"); assert_close(go, ro.float(), 1e-6, "scale_add d/do")
    assert_close(gs, rs.float(), 2e-5, "scale_add d/dscale")       # a sum of B H W products of both signs, fp32 accumulation
