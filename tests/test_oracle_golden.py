"""CPU: the C oracle against the golden vectors produced by the reference itself
(tests/golden/make_golden.py).  This is what pins the oracle."""
import pytest
import torch

from conftest import assert_close
from oracle import oracle


def test_dwt_iwt_bit_exact(golden):
    g = golden("wavelet")
    for tag in ("a", "b"):
        outs = oracle.dwt_raw(g[f"{tag}_x"])
        for name, o in zip(("ll", "hl", "lh", "hh"), outs):
            assert torch.equal(o, g[f"{tag}_{name}"]), f"{tag}_{name} not bit-exact"
        assert torch.equal(oracle.iwt_raw(g[f"{tag}_iwt_in"]), g[f"{tag}_iwt_out"])
        assert torch.equal(oracle.iwt_raw(torch.cat(outs, 1)), g[f"{tag}_rec"])


def test_dwt_odd_size_raises():
    with pytest.raises(RuntimeError):
        oracle.dwt_raw(torch.zeros(1, 1, 5, 4))


@pytest.mark.parametrize("tag", ["s16", "sq16", "s32", "d8"])
def test_scan_forward(golden, tag):
    g = golden("scan")
    y = oracle.selscan_fwd_raw(g[f"{tag}_u"], g[f"{tag}_delta"], g[f"{tag}_A"], g[f"{tag}_B"],
                               g[f"{tag}_C"], g[f"{tag}_D"], None, g[f"{tag}_bias"], True)
    assert_close(y, g[f"{tag}_y"], 1e-5, f"{tag} y")


@pytest.mark.parametrize("tag", ["s16", "sq16", "s32", "d8"])
def test_scan_backward(golden, tag):
    g = golden("scan")
    grads = oracle.selscan_bwd_raw(g[f"{tag}_u"], g[f"{tag}_delta"], g[f"{tag}_A"], g[f"{tag}_B"],
                                   g[f"{tag}_C"], g[f"{tag}_D"], g[f"{tag}_bias"], g[f"{tag}_dy"], True)
    for name, got in zip(("du", "ddelta", "dA", "dB", "dC", "dD", "dbias"), grads):
        assert_close(got, g[f"{tag}_{name}"], 2e-5, f"{tag} {name}")


def test_scan_optional_arguments(golden):
    g = golden("scan")
    y, last = oracle.selscan_fwd_raw(g["opt_u"], g["opt_delta"], g["opt_A"], g["opt_B"], g["opt_C"],
                                     g["opt_D"], g["opt_z"], g["opt_bias"], True, True)
    assert_close(y, g["opt_y_full"], 1e-5, "z-gated y")
    assert_close(last, g["opt_last_state"], 1e-5, "last state")
    y2 = oracle.selscan_fwd_raw(g["opt_u"], g["opt_delta"].abs() + 0.01, g["opt_A"], g["opt_B"],
                                g["opt_C"], None, None, None, False, False)
    assert_close(y2, g["opt_y_plain"], 1e-5, "plain y")


@pytest.mark.parametrize("tag", ["s16", "sq16", "s32", "d8"])
def test_ss2d_core_forward(golden, tag):
    g = golden("scan")
    ys = oracle.ss2d_core_raw(g[f"{tag}_core_x"], g[f"{tag}_x_proj_weight"], g[f"{tag}_dt_projs_weight"],
                              g[f"{tag}_dt_projs_bias"], g[f"{tag}_A_logs"], g[f"{tag}_Ds"])
    for i, y in enumerate(ys):
        assert_close(y, g[f"{tag}_core_y{i}"], 1e-5, f"{tag} core y{i}")
