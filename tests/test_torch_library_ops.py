"""The hot-path operators as torch.library custom ops (wave_mamba_amd/torch_ops.py).
CPU: registration, schemas and fake (meta) shapes - tracing never launches a kernel.  GPU: values and gradients equal
wave_mamba_amd.ops' (the same C ABI calls) and torch.library.opcheck accepts every op."""
import pytest
import torch

import wave_mamba_amd as wm
import wave_mamba_amd.torch_ops as tops


def test_ops_are_registered_with_fake_implementations():
    from torch._subclasses.fake_tensor import FakeTensorMode
    ns = torch.ops.wavemamba_hip
    assert all(hasattr(ns, n) for n in tops.OPS)
    assert "Tensor? D" in str(ns.selective_scan.default._schema) and "bool delta_softplus" in str(ns.selective_scan.default._schema)
    with FakeTensorMode():
        x = torch.empty(2, 8, 16, 24, device="cuda")
        outs = ns.dwt2d(x)
        assert [tuple(o.shape) for o in outs] == [(2, 8, 8, 12)] * 4 and outs[0].dtype == x.dtype
        y = ns.idwt2d(torch.cat(outs, 1).bfloat16())
        assert tuple(y.shape) == (2, 8, 16, 24) and y.dtype == torch.float32           # the reference's always-fp32 IWT (:122-123)
        assert tuple(ns.dwt2d_backward(*outs).shape) == (2, 8, 16, 24)
        assert ns.idwt2d_backward(y, True).dtype == torch.bfloat16
        u = torch.empty(2, 256, 640, device="cuda")
        A, Bm, D = torch.empty(256, 16, device="cuda"), torch.empty(2, 4, 16, 640, device="cuda"), torch.empty(256, device="cuda")
        assert tuple(ns.selective_scan(u, u, A, Bm, Bm, D, None, True).shape) == (2, 256, 640)
        g = ns.selective_scan_backward(u, u, A, Bm, Bm, D, None, u, True)
        assert [tuple(t.shape) for t in g] == [(2, 256, 640), (2, 256, 640), (256, 16), (2, 4, 16, 640), (2, 4, 16, 640), (256,), (0,)]
        xx = torch.empty(1, 64, 8, 12, device="cuda")
        p = [torch.empty(4, 34, 64, device="cuda"), torch.empty(4, 64, 2, device="cuda"), torch.empty(4, 64, device="cuda"),
             torch.empty(256, 16, device="cuda"), torch.empty(256, device="cuda")]
        ys = ns.ss2d_core(xx, *p)
        assert [tuple(t.shape) for t in ys] == [(1, 64, 96)] * 4
        assert [tuple(t.shape) for t in ns.ss2d_core_backward(xx, *p, *ys)] == [(1, 64, 8, 12)] + [tuple(t.shape) for t in p]
    with pytest.raises(Exception):
        with FakeTensorMode():
            ns.dwt2d(torch.empty(1, 1, 3, 4, device="cuda"))               # odd H: the reference raises (:97-110)


@pytest.mark.gpu
def test_library_ops_equal_the_python_surface_and_pass_opcheck():
    dev = "cuda:0"
    ns = torch.ops.wavemamba_hip
    g = torch.Generator().manual_seed(3)
    # DWT / IWT
    x = torch.randn(2, 6, 8, 12, generator=g).to(dev).requires_grad_(True)
    x2 = x.detach().clone().requires_grad_(True)
    a, b = ns.dwt2d(x), wm.ops.dwt_init(x2)
    assert all(torch.equal(p, q) for p, q in zip(a, b))
    ya, yb = ns.idwt2d(torch.cat(a, 1)), wm.ops.iwt_init(torch.cat(b, 1))
    assert torch.equal(ya, yb)
    w = torch.randn_like(ya)
    (ya * w).sum().backward(); (yb * w).sum().backward()
    assert torch.equal(x.grad, x2.grad)
    # selective scan
    from test_gpu_parity import random_scan_case, random_core_case
    args = [t.to(dev) for t in random_scan_case(2, 64, 200, 16, 4, seed=5)]
    l1 = [t.clone().requires_grad_(True) for t in args]
    l2 = [t.clone().requires_grad_(True) for t in args]
    o1 = ns.selective_scan(*l1, True)
    o2 = wm.ops.selective_scan_fn(l2[0], l2[1], l2[2], l2[3], l2[4], l2[5], None, l2[6], True)
    assert torch.equal(o1, o2)
    dy = torch.randn_like(o1)
    for ga, gb in zip(torch.autograd.grad(o1, l1, dy), torch.autograd.grad(o2, l2, dy)):
        assert torch.equal(ga, gb)
    # fused core
    c = [t.to(dev) for t in random_core_case(2, 64, 12, 20, 16, 2, seed=7)]
    c1 = [t.clone().requires_grad_(True) for t in c]
    c2 = [t.clone().requires_grad_(True) for t in c]
    y1, y2 = ns.ss2d_core(*c1), wm.ops.ss2d_core(*c2)
    assert all(torch.equal(p, q) for p, q in zip(y1, y2))
    dys = [torch.randn_like(t) for t in y1]
    for ga, gb in zip(torch.autograd.grad(y1, c1, dys), torch.autograd.grad(y2, c2, dys)):
        assert torch.equal(ga, gb)
    # the library's own consistency checks: fake implementation against the real one (shapes, dtypes, strides, device), and
    # that autograd is registered through torch.library (test_schema is left out: its aliasing probe flags the separately
    # allocated outputs of the multi-output ops)
    utils = ("test_faketensor", "test_autograd_registration")
    small = torch.randn(1, 2, 4, 8, device=dev, requires_grad=True)
    torch.library.opcheck(ns.dwt2d.default, (small,), test_utils=utils)
    torch.library.opcheck(ns.idwt2d.default, (torch.randn(1, 8, 4, 8, device=dev, requires_grad=True),), test_utils=utils)
    torch.library.opcheck(ns.selective_scan.default, tuple(l1) + (True,), test_utils=utils)
    torch.library.opcheck(ns.ss2d_core.default, tuple(c1), test_utils=utils)


@pytest.mark.gpu
def test_ops_trace_through_torch_compile_aot_eager():
    """A function built from the registered ops compiled with torch.compile(backend="aot_eager", fullgraph=True): Dynamo and AOT
    autograd trace forward AND backward through the fake implementations and the registered autograd formulas (the ops stay
    opaque nodes; no kernel runs while tracing), then the traced graphs run the HIP kernels - values and gradients equal eager's."""
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(3)
    ns = torch.ops.wavemamba_hip

    def f(x, u, delta, A, Bm, Cm, D, bias):
        ll, hl, lh, hh = ns.dwt2d(x)
        y = ns.idwt2d(torch.cat([ll * 1.5, hl, lh - hh, hh], 1))
        s = ns.selective_scan(u, delta, A, Bm, Cm, D, bias, True)
        return y.square().mean() + s.square().mean()

    x = torch.randn(2, 4, 16, 24, device=dev, generator=g)
    u, delta = torch.randn(2, 64, 200, device=dev, generator=g), torch.randn(2, 64, 200, device=dev, generator=g) * 0.3
    A = -torch.rand(64, 16, device=dev, generator=g) - 0.1
    Bm, Cm = torch.randn(2, 1, 16, 200, device=dev, generator=g), torch.randn(2, 1, 16, 200, device=dev, generator=g)
    D, bias = torch.randn(64, device=dev, generator=g), torch.randn(64, device=dev, generator=g) * 0.1

    def run(fn):
        ins = [t.clone().requires_grad_(True) for t in (x, u, delta, A, Bm, Cm, D, bias)]
        out = fn(*ins)
        out.backward()
        return out.detach(), [t.grad for t in ins]

    torch._dynamo.reset()
    o_e, g_e = run(f)
    o_c, g_c = run(torch.compile(f, backend="aot_eager", fullgraph=True))
    assert torch.equal(o_e, o_c)
    for a, b in zip(g_e, g_c):
        assert a is not None and b is not None and torch.equal(a, b)
