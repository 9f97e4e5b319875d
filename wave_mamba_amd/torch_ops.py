"""The hot-path operators as torch.library custom ops (namespace `wavemamba_hip`), for hosts that bind operators by name -
torch.compile / torch.export graphs, C++ frontends, other packages - instead of importing `wave_mamba_amd.ops`
("torch.library custom ops with registered backward named wavemamba_hip::{dwt2d, idwt2d, selective_scan,
ss2d_core}").

    import wave_mamba_amd.torch_ops            # registers the ops (idempotent)
    ll, hl, lh, hh = torch.ops.wavemamba_hip.dwt2d(x)                       # dwt_init          (wavemamba_arch.py:97-110)
    y = torch.ops.wavemamba_hip.idwt2d(torch.cat([ll, hl, lh, hh], 1))     # iwt_init          (:113-130)
    out = torch.ops.wavemamba_hip.selective_scan(u, delta, A, B, C, D, delta_bias, True)   # selective_scan_fn (:465-471)
    y0, y1, y2, y3 = torch.ops.wavemamba_hip.ss2d_core(x, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds)  # (:446-478)

Every op is a thin shell over the C ABI calls `ops.py` makes (same kernels, same checks) under the CUDA dispatch key, has a
fake (meta) implementation so that tracing never launches anything, and an autograd formula whose backward is itself a
registered op (`..._backward`), so compiled / exported graphs hold both directions as opaque nodes.  Under the CPU dispatch key
the ops are the plain-PyTorch twins of cpu_twin.py ("each with CPU + HIP implementations"; their backward ops
differentiate the twin with autograd) - the dispatcher picks by the tensors' device, a CUDA tensor never reaches them.
"""
import torch

from . import cpu_twin, ops

_NS = "wavemamba_hip"
_lib_def = torch.library.Library(_NS, "FRAGMENT")


class _Ctx:
    """What the autograd.Function bodies of ops.py store on `ctx`, for calling them outside autograd."""

    def __init__(self):
        self.saved_tensors = ()

    def save_for_backward(self, *t):
        self.saved_tensors = t

    def mark_non_differentiable(self, *t):
        pass


def _define(name, schema, impl, fake):
    _lib_def.define(f"{name}{schema}")
    torch.library.impl(f"{_NS}::{name}", "CUDA", lib=_lib_def)(impl)
    torch.library.register_fake(f"{_NS}::{name}", fake, lib=_lib_def)


# ---- Haar DWT / IWT -------------------------------------------------------------------------------------------------
def _dwt2d(x):
    return ops._DWT.forward(_Ctx(), x)


def _dwt2d_fake(x):
    B, C, H, W = x.shape
    torch._check(H % 2 == 0 and W % 2 == 0, lambda: f"dwt_init: H and W must be even, got {H}x{W}")
    return tuple(x.new_empty((B, C, H // 2, W // 2)) for _ in range(4))


def _dwt2d_backward(g_ll, g_hl, g_lh, g_hh):
    ctx = _Ctx()
    B, C, h, w = g_ll.shape
    ctx.shape = (B, C, 2 * h, 2 * w)
    return ops._DWT.backward(ctx, g_ll, g_hl, g_lh, g_hh)


def _dwt2d_backward_fake(g_ll, g_hl, g_lh, g_hh):
    B, C, h, w = g_ll.shape
    return g_ll.new_empty((B, C, 2 * h, 2 * w))


def _idwt2d(x):
    return ops._IWT.forward(_Ctx(), x, None)


def _idwt2d_fake(x):
    B, C4, h, w = x.shape
    torch._check(C4 % 4 == 0, lambda: f"iwt_init: channel count {C4} is not a multiple of 4")
    return x.new_empty((B, C4 // 4, 2 * h, 2 * w), dtype=torch.float32)          # always fp32 (:122-123)


def _idwt2d_backward(g, bf16):
    ctx = _Ctx()
    B, C, H, W = g.shape
    ctx.geom = (B, C, H // 2, W // 2, True, torch.bfloat16 if bf16 else torch.float32)
    return ops._IWT.backward(ctx, g)[0]


def _idwt2d_backward_fake(g, bf16):
    B, C, H, W = g.shape
    return g.new_empty((B, 4 * C, H // 2, W // 2), dtype=torch.bfloat16 if bf16 else torch.float32)


# ---- selective scan -------------------------------------------------------------------------------------------------
def _selective_scan(u, delta, A, B, C, D, delta_bias, delta_softplus):
    ops._scan_shapes(u, delta, A, B, C, D, None, delta_bias)
    args = [ops._f32c(t) for t in (u, delta, A, B if B.dim() == 4 else B.unsqueeze(1), C if C.dim() == 4 else C.unsqueeze(1),
                                   D, delta_bias)]
    out, _ = ops._scan_forward(args[0], args[1], args[2], args[3], args[4], args[5], None, args[6], delta_softplus, False)
    return out.to(u.dtype)


def _selective_scan_fake(u, delta, A, B, C, D, delta_bias, delta_softplus):
    return torch.empty_like(u)


def _selective_scan_backward(u, delta, A, B, C, D, delta_bias, dout, delta_softplus):
    ctx = _Ctx()
    B4, C4 = (B if B.dim() == 4 else B.unsqueeze(1)), (C if C.dim() == 4 else C.unsqueeze(1))
    ctx.saved_tensors = tuple(ops._f32c(t) for t in (u, delta, A, B4, C4, D, delta_bias))
    ctx.delta_softplus = bool(delta_softplus)
    du, ddelta, dA, dB, dC, dD, dbias, _, _ = ops._SelectiveScan.backward(ctx, dout)
    z = u.new_zeros((0,), dtype=torch.float32)
    return (du.to(u.dtype), ddelta.to(delta.dtype), dA, dB.reshape(B.shape), dC.reshape(C.shape), z if dD is None else dD,
            z if dbias is None else dbias)


def _selective_scan_backward_fake(u, delta, A, B, C, D, delta_bias, dout, delta_softplus):
    z = u.new_empty((0,), dtype=torch.float32)
    f = lambda t: t.new_empty(t.shape, dtype=torch.float32)
    return (torch.empty_like(u), torch.empty_like(delta), f(A), f(B), f(C), z if D is None else f(D),
            z if delta_bias is None else f(delta_bias))


# ---- fused SS2D core --------------------------------------------------------------------------------------------------
def _ss2d_core(x, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds):
    f = [x.detach().contiguous().float()] + [t.detach().contiguous().float()
                                               for t in (x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds)]
    return tuple(ops._ss2d_core_fwd(f, False, separate=True))


def _ss2d_core_fake(x, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds):
    B, D, H, W = x.shape
    return tuple(x.new_empty((B, D, H * W), dtype=torch.float32) for _ in range(4))


def _ss2d_core_backward(x, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds, dy0, dy1, dy2, dy3):
    ctx = _Ctx()
    ctx.saved_tensors = tuple(t.detach().contiguous().float() for t in (x, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds))
    ctx.merged = False
    return tuple(ops._SS2DCoreFn.backward(ctx, dy0, dy1, dy2, dy3)[1:])


def _ss2d_core_backward_fake(x, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds, dy0, dy1, dy2, dy3):
    return tuple(t.new_empty(t.shape, dtype=torch.float32) for t in (x, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds))


_define("dwt2d", "(Tensor x) -> (Tensor, Tensor, Tensor, Tensor)", _dwt2d, _dwt2d_fake)
_define("dwt2d_backward", "(Tensor g_ll, Tensor g_hl, Tensor g_lh, Tensor g_hh) -> Tensor", _dwt2d_backward, _dwt2d_backward_fake)
_define("idwt2d", "(Tensor x) -> Tensor", _idwt2d, _idwt2d_fake)
_define("idwt2d_backward", "(Tensor g, bool bf16) -> Tensor", _idwt2d_backward, _idwt2d_backward_fake)
_define("selective_scan", "(Tensor u, Tensor delta, Tensor A, Tensor B, Tensor C, Tensor? D, Tensor? delta_bias, "
        "bool delta_softplus) -> Tensor", _selective_scan, _selective_scan_fake)
_define("selective_scan_backward", "(Tensor u, Tensor delta, Tensor A, Tensor B, Tensor C, Tensor? D, Tensor? delta_bias, "
        "Tensor dout, bool delta_softplus) -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)",
        _selective_scan_backward, _selective_scan_backward_fake)
_define("ss2d_core", "(Tensor x, Tensor x_proj_weight, Tensor dt_projs_weight, Tensor dt_projs_bias, Tensor A_logs, Tensor Ds)"
        " -> (Tensor, Tensor, Tensor, Tensor)", _ss2d_core, _ss2d_core_fake)
_define("ss2d_core_backward", "(Tensor x, Tensor x_proj_weight, Tensor dt_projs_weight, Tensor dt_projs_bias, Tensor A_logs, "
        "Tensor Ds, Tensor dy0, Tensor dy1, Tensor dy2, Tensor dy3) -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)",
        _ss2d_core_backward, _ss2d_core_backward_fake)



# ---- CPU dispatch key: the plain-PyTorch twins --------------------------------------------------------------------------
def _twin_vjp(fn, inputs, grad_outputs):
    """Gradients of the twin `fn` at `inputs` (None entries: absent optional arguments) for the output cotangents."""
    with torch.enable_grad():
        leaves = [None if t is None else t.detach().clone().requires_grad_(True) for t in inputs]
        outs = fn(*leaves)
        outs = outs if isinstance(outs, tuple) else (outs,)
        have = [l for l in leaves if l is not None]
        gs = list(torch.autograd.grad(outs, have, grad_outputs, allow_unused=True))
    return [None if l is None else (torch.zeros_like(l) if (g := gs.pop(0)) is None else g) for l in leaves]


def _cpu(name, impl):
    torch.library.impl(f"{_NS}::{name}", "CPU", lib=_lib_def)(impl)


_cpu("dwt2d", lambda x: tuple(t.contiguous() for t in cpu_twin.dwt_init(x)))
_cpu("dwt2d_backward", lambda g_ll, g_hl, g_lh, g_hh: _twin_vjp(
    cpu_twin.dwt_init, [g_ll.new_zeros(g_ll.shape[0], g_ll.shape[1], 2 * g_ll.shape[2], 2 * g_ll.shape[3])],
    [g_ll, g_hl, g_lh, g_hh])[0])
_cpu("idwt2d", cpu_twin.iwt_init)
_cpu("idwt2d_backward", lambda g, bf16: _twin_vjp(
    cpu_twin.iwt_init, [g.new_zeros(g.shape[0], 4 * g.shape[1], g.shape[2] // 2, g.shape[3] // 2,
                                    dtype=torch.bfloat16 if bf16 else torch.float32)], [g])[0])
_cpu("selective_scan", lambda u, delta, A, B, C, D, delta_bias, delta_softplus: cpu_twin.selective_scan_fn(
    u, delta, A, B, C, D, None, delta_bias, delta_softplus))


def _selective_scan_backward_cpu(u, delta, A, B, C, D, delta_bias, dout, delta_softplus):
    gs = _twin_vjp(lambda u_, d_, A_, B_, C_, D_, b_: cpu_twin.selective_scan_fn(u_, d_, A_, B_, C_, D_, None, b_, delta_softplus),
                   [u, delta, A, B, C, D, delta_bias], [dout])
    z = u.new_zeros((0,), dtype=torch.float32)
    return (gs[0], gs[1], gs[2].float(), gs[3].float(), gs[4].float(), z if gs[5] is None else gs[5].float(),
            z if gs[6] is None else gs[6].float())


_cpu("selective_scan_backward", _selective_scan_backward_cpu)
_cpu("ss2d_core", lambda x, Wx, Wdt, bias, A_logs, Ds: tuple(t.contiguous() for t in cpu_twin.ss2d_core(x, Wx, Wdt, bias, A_logs, Ds)))
_cpu("ss2d_core_backward", lambda x, Wx, Wdt, bias, A_logs, Ds, dy0, dy1, dy2, dy3: tuple(
    g.float() for g in _twin_vjp(cpu_twin.ss2d_core, [x, Wx, Wdt, bias, A_logs, Ds], [dy0, dy1, dy2, dy3])))

_o = torch.ops.wavemamba_hip


# ---- autograd formulas (the backward of every op is another registered op) ----------------------------------------------
def _dwt_setup(ctx, inputs, output):
    ctx.meta = (output[0].shape, output[0].dtype, output[0].device)


def _dwt_backward(ctx, g0, g1, g2, g3):
    shape, dtype, device = ctx.meta
    gs = [torch.zeros(shape, dtype=dtype, device=device) if g is None else g.contiguous() for g in (g0, g1, g2, g3)]
    return _o.dwt2d_backward(*gs)


torch.library.register_autograd(f"{_NS}::dwt2d", _dwt_backward, setup_context=_dwt_setup, lib=_lib_def)


def _idwt_setup(ctx, inputs, output):
    ctx.bf16 = inputs[0].dtype == torch.bfloat16


torch.library.register_autograd(f"{_NS}::idwt2d", lambda ctx, g: _o.idwt2d_backward(g.contiguous().float(), ctx.bf16),
                                setup_context=_idwt_setup, lib=_lib_def)


def _scan_setup(ctx, inputs, output):
    u, delta, A, B, C, D, bias, sp = inputs
    ctx.has = (D is not None, bias is not None)
    ctx.sp = sp
    ctx.save_for_backward(*[t for t in (u, delta, A, B, C, D, bias) if t is not None])


def _scan_backward(ctx, dout):
    s = list(ctx.saved_tensors)
    u, delta, A, B, C = s[:5]
    rest = s[5:]
    D = rest.pop(0) if ctx.has[0] else None
    bias = rest.pop(0) if ctx.has[1] else None
    du, dd, dA, dB, dC, dD, db = _o.selective_scan_backward(u, delta, A, B, C, D, bias, dout.contiguous(), ctx.sp)
    return du, dd, dA, dB, dC, (dD if ctx.has[0] else None), (db if ctx.has[1] else None), None


torch.library.register_autograd(f"{_NS}::selective_scan", _scan_backward, setup_context=_scan_setup, lib=_lib_def)


def _core_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)
    ctx.out_shape = output[0].shape


def _core_backward(ctx, g0, g1, g2, g3):
    x = ctx.saved_tensors[0]
    gs = [torch.zeros(ctx.out_shape, dtype=torch.float32, device=x.device) if g is None else g.contiguous().float()
          for g in (g0, g1, g2, g3)]
    return _o.ss2d_core_backward(*ctx.saved_tensors, *gs)


torch.library.register_autograd(f"{_NS}::ss2d_core", _core_backward, setup_context=_core_setup, lib=_lib_def)

OPS = ("dwt2d", "dwt2d_backward", "idwt2d", "idwt2d_backward", "selective_scan", "selective_scan_backward", "ss2d_core",
       "ss2d_core_backward")
