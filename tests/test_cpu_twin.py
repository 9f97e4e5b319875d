"""CPU: the package's own CPU twins of the three boundary operators (wave_mamba_amd/cpu_twin.py, plain PyTorch) against the
golden vectors the reference itself produced (tests/golden/make_golden.py), and BASELINE config 1 - the network's "CPU-only
PyTorch forward" - on them, WITHOUT the test oracle installed.  The twins serve CPU tensors only; the last tests pin that no other
device reaches them and that nothing else in `ops` accepts a CPU tensor."""
import pytest
import torch

from conftest import assert_close
import wave_mamba_amd as wm
from wave_mamba_amd import cpu_twin

SHIPPED = dict(in_chn=3, wf=32, n_l_blocks=[1, 2, 4], n_h_blocks=[1, 1, 2], ffn_scale=2.0)


def gen(seed):
    g = torch.Generator()
    g.manual_seed(seed)
    return g


def test_dwt_iwt_twins_bit_exact_with_the_reference(golden):
    g = golden("wavelet")
    for tag in ("a", "b"):
        outs = wm.ops.dwt_init(g[f"{tag}_x"])                               # through the product's dispatch: CPU tensor -> twin
        for name, o in zip(("ll", "hl", "lh", "hh"), outs):
            assert torch.equal(o, g[f"{tag}_{name}"]), f"{tag}_{name} not bit-exact"
        assert torch.equal(wm.ops.iwt_init(g[f"{tag}_iwt_in"]), g[f"{tag}_iwt_out"])
        assert torch.equal(wm.ops.iwt_init(torch.cat(outs, 1)), g[f"{tag}_rec"])
        assert torch.equal(wm.ops.iwt_init_pair(outs[0], torch.cat(outs[1:], 1)), g[f"{tag}_rec"])
    with pytest.raises(RuntimeError):
        wm.ops.dwt_init(torch.zeros(1, 1, 5, 4))                            # odd size: as the reference (RuntimeError)
    assert wm.ops.iwt_init(torch.zeros(1, 4, 2, 2, dtype=torch.bfloat16)).dtype == torch.float32     # the fp32-out quirk (:122-123)
    assert wm.ops.dwt_init(torch.zeros(1, 1, 2, 2, dtype=torch.bfloat16))[0].dtype == torch.bfloat16


@pytest.mark.parametrize("tag", ["s16", "sq16", "s32", "d8"])
def test_scan_twin_forward_and_backward(golden, tag):
    g = golden("scan")
    names = ("u", "delta", "A", "B", "C", "D", "bias")
    args = [g[f"{tag}_{n}"].clone().requires_grad_(True) for n in names]
    y = wm.ops.selective_scan_fn(args[0], args[1], args[2], args[3], args[4], args[5], None, args[6], True)
    assert_close(y, g[f"{tag}_y"], 1e-5, f"{tag} y")
    grads = torch.autograd.grad(y, args, g[f"{tag}_dy"])
    for name, got in zip(("du", "ddelta", "dA", "dB", "dC", "dD", "dbias"), grads):
        assert_close(got, g[f"{tag}_{name}"], 2e-5, f"{tag} {name}")


def test_scan_twin_optional_arguments(golden):
    g = golden("scan")
    y, last = wm.ops.selective_scan_fn(g["opt_u"], g["opt_delta"], g["opt_A"], g["opt_B"], g["opt_C"], g["opt_D"], g["opt_z"],
                                       g["opt_bias"], True, True)
    assert_close(y, g["opt_y_full"], 1e-5, "z-gated y")
    assert_close(last, g["opt_last_state"], 1e-5, "last state")
    y2 = wm.ops.selective_scan_fn(g["opt_u"], g["opt_delta"].abs() + 0.01, g["opt_A"], g["opt_B"], g["opt_C"])
    assert_close(y2, g["opt_y_plain"], 1e-5, "plain y")
    empty = wm.ops.selective_scan_fn(torch.zeros(1, 4, 0), torch.zeros(1, 4, 0), -torch.ones(4, 2), torch.zeros(1, 1, 2, 0),
                                     torch.zeros(1, 1, 2, 0))
    assert empty.shape == (1, 4, 0)
    with pytest.raises(RuntimeError):
        wm.ops.selective_scan_fn(torch.zeros(1, 4, 8), torch.zeros(1, 4, 7), -torch.ones(4, 2), torch.zeros(1, 1, 2, 8),
                                 torch.zeros(1, 1, 2, 8))


@pytest.mark.parametrize("tag", ["s16", "sq16", "s32", "d8"])
def test_forward_core_on_the_twins(golden, tag):
    """SS2D.forward_core's direction glue (the arch's own, :446-478) over the scan twin against the reference's forward_core."""
    from wave_mamba_amd.archs import wavemamba_arch as arch
    g = golden("scan")
    x = g[f"{tag}_core_x"]
    D, N, R = x.shape[1], g[f"{tag}_A_logs"].shape[1], g[f"{tag}_dt_projs_weight"].shape[2]
    ss = arch.SS2D(d_model=D // 2, d_state=N, expand=2.0).eval()
    assert ss.dt_rank == R and ss.d_inner == D
    with torch.no_grad():
        for name in ("x_proj_weight", "dt_projs_weight", "dt_projs_bias", "A_logs", "Ds"):
            getattr(ss, name).copy_(g[f"{tag}_{name}"])
        ys = ss.forward_core(x)
    for i, y in enumerate(ys):
        assert_close(y, g[f"{tag}_core_y{i}"], 1e-5, f"{tag} core y{i}")


@pytest.mark.parametrize("tag,hw", [("32x64", (32, 64)), ("128x128", (128, 128)), ("256x256", (256, 256))])
def test_config1_cpu_forward_on_the_twins(golden, tag, hw):
    """BASELINE config 1 (1 x 3 x 256 x 256, CPU-only forward) and the two smaller pins: the shipped network on CPU tensors, product
    operators only (no test backend installed), against the reference's outputs."""
    from wave_mamba_amd.archs import wavemamba_arch as arch
    assert arch._OpsBackend.impl is wm.ops
    torch.manual_seed(0)
    net = wm.WaveMamba(**SHIPPED).eval()
    x = torch.rand(1, 3, *hw, generator=gen(1234))
    with torch.no_grad():
        y = net.restoration_network(x)
    assert_close(y, golden("model_shipped")[f"y_{tag}"], 1e-5, f"shipped {tag} on the CPU twins")


def test_other_devices_never_reach_the_cpu_twins(monkeypatch):
    """The dispatch key is `device.type == "cpu"` for EVERY tensor argument: a tensor of any other device (here: meta - this test
    has no GPU) and mixed arguments take the HIP path, which refuses what is not a CUDA tensor.  The GPU twin of this test
    (tests/test_gpu_parity.py::test_cuda_tensors_fail_loudly_without_the_library) removes the library under a CUDA tensor."""
    def boom(*a, **k):
        raise AssertionError("the CPU twin was reached by a non-CPU tensor")
    for name in ("dwt_init", "iwt_init", "iwt_init_pair", "selective_scan_fn"):
        monkeypatch.setattr(cpu_twin, name, boom)
    m = torch.zeros(1, 4, 4, 4, device="meta")
    for call in (lambda: wm.ops.dwt_init(m), lambda: wm.ops.iwt_init(m), lambda: wm.ops.iwt_init_pair(m[:, :1], m[:, 1:]),
                 lambda: wm.ops.iwt_init_pair(torch.zeros(1, 1, 4, 4), m[:, 1:]),
                 lambda: wm.ops.selective_scan_fn(torch.zeros(1, 4, 8, device="meta"), torch.zeros(1, 4, 8), -torch.ones(4, 2),
                                                  torch.zeros(1, 1, 2, 8), torch.zeros(1, 1, 2, 8))):
        with pytest.raises(RuntimeError) as e:
            call()
        assert "CPU twin was reached" not in str(e.value)


def test_the_rest_of_ops_refuses_cpu_tensors():
    """Only the three boundary operators have a CPU twin; the fused kernels' wrappers have no CPU path at all."""
    x = torch.zeros(1, 64, 8, 8)
    with pytest.raises(RuntimeError):
        wm.ops.ss2d_core(x, torch.zeros(4, 34, 64), torch.zeros(4, 64, 2), torch.zeros(4, 64), torch.zeros(256, 16), torch.ones(256))
    with pytest.raises(RuntimeError):
        wm.ops.conv2d(x, torch.zeros(64, 64, 3, 3))
    with pytest.raises(RuntimeError):
        wm.ops.dwconv3x3(x, torch.zeros(64, 1, 3, 3), None, "none")


def test_torch_library_ops_on_cpu_tensors(golden):
    """wavemamba_hip::{dwt2d, idwt2d, selective_scan, ss2d_core} under the CPU dispatch key (wave_mamba_amd/torch_ops.py): forward
    and registered backward against the reference's goldens."""
    import wave_mamba_amd.torch_ops  # noqa: F401  (registers the ops)
    o = torch.ops.wavemamba_hip
    w = golden("wavelet")
    x = w["a_x"].clone().requires_grad_(True)
    outs = o.dwt2d(x)
    for name, t in zip(("ll", "hl", "lh", "hh"), outs):
        assert torch.equal(t, w[f"a_{name}"])
    rec = o.idwt2d(torch.cat(outs, 1))
    assert torch.equal(rec.detach(), w["a_rec"])
    gx, = torch.autograd.grad(rec, x, torch.ones_like(rec))                  # iwt(dwt(x)) = x: the chain's Jacobian is the identity
    assert_close(gx, torch.ones_like(x), 1e-6, "d iwt(dwt(x)) / dx")
    g = golden("scan")
    tag = "s16"
    names = ("u", "delta", "A", "B", "C", "D", "bias")
    args = [g[f"{tag}_{n}"].clone().requires_grad_(True) for n in names]
    y = o.selective_scan(*args, True)
    assert_close(y, g[f"{tag}_y"], 1e-5, "selective_scan on CPU")
    grads = torch.autograd.grad(y, args, g[f"{tag}_dy"])
    for name, got in zip(("du", "ddelta", "dA", "dB", "dC", "dD", "dbias"), grads):
        assert_close(got, g[f"{tag}_{name}"], 2e-5, f"selective_scan backward on CPU: {name}")
    pars = [g[f"{tag}_{n}"].clone().requires_grad_(True) for n in ("core_x", "x_proj_weight", "dt_projs_weight", "dt_projs_bias", "A_logs", "Ds")]
    ys = o.ss2d_core(*pars)
    for i, yk in enumerate(ys):
        assert_close(yk, g[f"{tag}_core_y{i}"], 1e-5, f"ss2d_core on CPU: y{i}")
    gs = torch.autograd.grad(sum(yk.sum() for yk in ys), pars)
    assert all(t.shape == p.shape and bool(torch.isfinite(t).all()) for t, p in zip(gs, pars))
