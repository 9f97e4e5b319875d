"""Host-side mirror of the reference's operator surface for the Wave-Mamba hot path.

Same names, argument meaning and error behaviour as the reference
(/root/reference/basicsr/archs/wavemamba_arch.py):

    dwt_init(x) -> (x_LL, x_HL, x_LH, x_HH)                                   :97-110
    iwt_init(x) -> h                                                          :113-130
    selective_scan_fn(u, delta, A, B, C, D=None, z=None, delta_bias=None,
                      delta_softplus=False, return_last_state=False)          :6, :465-471

Every op is a torch.autograd.Function over the C ABI of libwavemamba_hip.so (hand-written HIP for
gfx950).  PyTorch only supplies device memory, the current stream and autograd plumbing.  There is
NO CPU / eager fallback: a non-CUDA tensor or a missing library raises.
"""
import os

import torch
import torch.nn.functional as F

from . import _lib
from ._lib import WM_BF16, WM_F32, check


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _require_cuda(name, *tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                f"{name}: expected a CUDA (ROCm) tensor, got device '{t.device}'. The Wave-Mamba hot "
                "path is HIP-only (dwt_init / iwt_init / selective_scan_fn alone have a CPU twin, for inputs that are ALL "
                "on the CPU); nothing falls back to the CPU.")


def _dtype_code(t, name):
    if t.dtype == torch.float32:
        return WM_F32
    if t.dtype == torch.bfloat16:
        return WM_BF16
    raise RuntimeError(f"{name}: unsupported dtype {t.dtype} (float32 and bfloat16 are implemented)")


def _ptr(t):
    return None if t is None else t.data_ptr()


# ------------------------------------------------------------------------------------------------
# Zeroed accumulator outputs (training): parameter gradients the kernels add into with atomics and the running maxima of the
# fp16-split convolutions are a few hundred bytes each, and the library's memset node in front of each of them was 4.2 us of
# stream time, 323 times per BASELINE config-3 step.  They are handed out of a 4-MB block zeroed ONCE (per device and stream,
# bump allocation, no slice is ever handed out twice); the block is registered with the library (wm_zero_arena_register), which
# then skips its memset for buffers inside it.  A block lives as long as any slice of it does (the slices are views of its
# storage); it is unregistered when the next block replaces it.
# ------------------------------------------------------------------------------------------------
# ONE-SHOT RULE: a slice is zero exactly once - when it is handed out.  Whoever receives it (a library call that accumulates into
# it, then autograd, which may keep it as a parameter's .grad) must never pass it to an accumulate-into entry point again: the
# library would take it for zero (it lies in a registered range) and add onto the old contents.  Nothing in this package does;
# a caller that keeps such tensors and wants to be safe calls zero_arena_clear() (new slices then come from a fresh block, and
# the old block, once its last slice is gone, is unregistered and freed).
_ZERO_ARENA_FLOATS = 1 << 20
_ZERO_ARENAS = {}            # (device index, stream handle) -> [block, floats handed out]
_ZERO_ARENA_LOCK = __import__("threading").Lock()


def _zeros_small(n, device):
    """A zeroed float32 vector of n elements on `device` for an accumulate-into output of the library, valid on the current
    stream.  Larger than 64 KB, or under graph capture: uninitialised memory (the library zeroes it itself)."""
    if n > 16384 or torch.cuda.is_current_stream_capturing():
        return torch.empty(n, dtype=torch.float32, device=device)
    import weakref
    lib = _lib.load()
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)
    step = (n + 63) & ~63                                  # 256-byte slots
    with _ZERO_ARENA_LOCK:
        ent = _ZERO_ARENAS.get(key)
        if ent is None or ent[1] + step > _ZERO_ARENA_FLOATS:
            block = torch.zeros(_ZERO_ARENA_FLOATS, dtype=torch.float32, device=device)
            if ent is not None:
                ent[2]()                                   # the exhausted block: unregistered now, freed with its last slice
            check(lib.wm_zero_arena_register(block.data_ptr(), 4 * _ZERO_ARENA_FLOATS), "wm_zero_arena_register")
            ent = _ZERO_ARENAS[key] = [block, 0, weakref.finalize(block, lib.wm_zero_arena_unregister, block.data_ptr())]
        out = ent[0][ent[1]:ent[1] + n]
        ent[1] += step
    return out


def zero_arena_clear():
    """Drop every zeroed arena block (all devices, all streams): each is unregistered from the library now and freed with its last
    slice.  For long-running processes that create and destroy streams (an entry per (device, stream) would otherwise keep a 4-MB
    block registered for ever - ADVICE r5), and for callers that re-use gradient tensors handed out of an arena (see the one-shot
    rule above).  Must not run concurrently with training steps on other threads (their next small output simply opens a new block)."""
    with _ZERO_ARENA_LOCK:
        for ent in _ZERO_ARENAS.values():
            ent[2]()                                       # weakref.finalize: unregister (idempotent)
        _ZERO_ARENAS.clear()


# ------------------------------------------------------------------------------------------------
# bf16-storage mode (BASELINE config 2 as worded): the plane tensors an LFSSBlock hands from kernel to kernel - x, z,
# the conv outputs, the four scan outputs, f, fc - are stored as bfloat16; every kernel computes in fp32 (tiles,
# projections, scan state, LayerNorm statistics) and the token tensors between blocks stay fp32.  Off by default:
# the parity bars of the fp32 path (1e-4 per tensor) cannot hold at 8 mantissa bits; bench.py --bf16 and
# tests/test_gpu_parity.py report the PSNR against the fp32 output.
# ------------------------------------------------------------------------------------------------
_PLANE_DTYPE = torch.float32


def set_plane_dtype(dtype):
    """torch.float32 (default) or torch.bfloat16: storage of the LFSSBlock-internal plane tensors on the fused inference
    path (shipped width C = 32 only).  Returns the previous setting."""
    global _PLANE_DTYPE
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("plane dtype must be torch.float32 or torch.bfloat16")
    prev, _PLANE_DTYPE = _PLANE_DTYPE, dtype
    return prev


def get_plane_dtype():
    return _PLANE_DTYPE


# ------------------------------------------------------------------------------------------------
# Haar DWT / IWT
# ------------------------------------------------------------------------------------------------
class _DWT(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        _require_cuda("dwt_init", x)
        if x.dim() != 4:
            raise RuntimeError(f"dwt_init: expected a 4-D NCHW tensor, got {tuple(x.shape)}")
        B, C, H, W = x.shape
        if H % 2 or W % 2:
            # the reference's strided slices disagree in size for odd H/W -> RuntimeError
            raise RuntimeError(f"dwt_init: H and W must be even, got {H}x{W}")
        code = _dtype_code(x, "dwt_init")
        x = x.contiguous()
        outs = [torch.empty((B, C, H // 2, W // 2), dtype=x.dtype, device=x.device) for _ in range(4)]
        with torch.cuda.device(x.device):
            check(lib.wm_dwt2d_fwd(_ptr(x), *[_ptr(o) for o in outs], B, C, H, W, code, _stream()),
                  "wm_dwt2d_fwd")
        ctx.shape = (B, C, H, W)
        return tuple(outs)

    @staticmethod
    def backward(ctx, g_ll, g_hl, g_lh, g_hh):
        lib = _lib.load()
        B, C, H, W = ctx.shape
        gs = [g.contiguous() for g in (g_ll, g_hl, g_lh, g_hh)]
        code = _dtype_code(gs[0], "dwt_init.backward")
        dx = torch.empty((B, C, H, W), dtype=gs[0].dtype, device=gs[0].device)
        with torch.cuda.device(dx.device):
            check(lib.wm_dwt2d_bwd(*[_ptr(g) for g in gs], _ptr(dx), B, C, H, W, code, _stream()),
                  "wm_dwt2d_bwd")
        return dx


class _IWT(torch.autograd.Function):
    """Synthesis from four sub-band blocks given as (tensor, channel offset) views of 1 or 2 tensors."""

    @staticmethod
    def forward(ctx, x_l, x_h):
        # x_h is None: x_l is the reference's concatenated (B, 4C, h, w) tensor.
        # else: x_l (B, C, h, w) and x_h (B, 3C, h, w) - upFRG's un-concatenated pair (:1006).
        lib = _lib.load()
        _require_cuda("iwt_init", x_l, x_h)
        x_l = x_l.contiguous()
        if x_h is None:
            B, C4, h, w = x_l.shape
            if C4 % 4:
                raise RuntimeError(f"iwt_init: channel count {C4} is not a multiple of 4")
            C = C4 // 4
            hw = h * w
            es = x_l.element_size()
            ptrs = [x_l.data_ptr() + k * C * hw * es for k in range(4)]
            strides = [4 * C * hw] * 4
            code = _dtype_code(x_l, "iwt_init")
        else:
            x_h = x_h.contiguous()
            if x_h.dtype != x_l.dtype:           # mixed precision (autocast): the output is fp32 anyway
                x_l, x_h = x_l.float(), x_h.float()
            B, C, h, w = x_l.shape
            if x_h.shape != (B, 3 * C, h, w):
                raise RuntimeError(f"iwt_init: x_h shape {tuple(x_h.shape)} != {(B, 3 * C, h, w)}")
            hw = h * w
            es = x_h.element_size()
            ptrs = [x_l.data_ptr()] + [x_h.data_ptr() + k * C * hw * es for k in range(3)]
            strides = [C * hw] + [3 * C * hw] * 3
            code = _dtype_code(x_l, "iwt_init")
        out = torch.empty((B, C, 2 * h, 2 * w), dtype=torch.float32, device=x_l.device)  # always fp32
        with torch.cuda.device(out.device):
            check(lib.wm_idwt2d_fwd(*ptrs, *strides, _ptr(out), B, C, h, w, code, _stream()), "wm_idwt2d_fwd")
        ctx.geom = (B, C, h, w, x_h is None, x_l.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        B, C, h, w, cat, dt = ctx.geom
        g = g.contiguous().float()
        hw = h * w
        code = WM_F32 if dt == torch.float32 else WM_BF16
        if cat:
            d_l = torch.empty((B, 4 * C, h, w), dtype=dt, device=g.device)
            d_h = None
            es = d_l.element_size()
            ptrs = [d_l.data_ptr() + k * C * hw * es for k in range(4)]
            strides = [4 * C * hw] * 4
        else:
            d_l = torch.empty((B, C, h, w), dtype=dt, device=g.device)
            d_h = torch.empty((B, 3 * C, h, w), dtype=dt, device=g.device)
            es = d_l.element_size()
            ptrs = [d_l.data_ptr()] + [d_h.data_ptr() + k * C * hw * es for k in range(3)]
            strides = [C * hw] + [3 * C * hw] * 3
        with torch.cuda.device(g.device):
            check(lib.wm_idwt2d_bwd(_ptr(g), *ptrs, *strides, B, C, h, w, code, _stream()), "wm_idwt2d_bwd")
        return d_l, d_h


# The three operators of the drop-in boundary have a CPU twin (cpu_twin.py: plain PyTorch, SURVEY.md 8b "each with CPU + HIP
# implementations", BASELINE config 1 "CPU-only PyTorch forward").  The dispatch key is the DEVICE OF THE INPUT, as for any torch
# operator - never the presence of the library: a CUDA tensor goes to the HIP kernels below and raises if they cannot be loaded.
def _on_cpu(*tensors):
    ts = [t for t in tensors if t is not None]
    return bool(ts) and all(t.device.type == "cpu" for t in ts)


def dwt_init(x):
    """Haar analysis of an NCHW map -> (x_LL, x_HL, x_LH, x_HH), each (B, C, H/2, W/2), dtype of x."""
    if _on_cpu(x):
        from . import cpu_twin
        return cpu_twin.dwt_init(x)
    return _DWT.apply(x)


def iwt_init(x):
    """Haar synthesis of a (B, 4C, h, w) tensor [x1|x2|x3|x4] -> (B, C, 2h, 2w), always float32."""
    if _on_cpu(x):
        from . import cpu_twin
        return cpu_twin.iwt_init(x)
    return _IWT.apply(x, None)


def iwt_init_pair(x_l, x_h):
    """iwt_init(torch.cat([x_l, x_h], dim=1)) without materialising the concatenation."""
    if _on_cpu(x_l, x_h):
        from . import cpu_twin
        return cpu_twin.iwt_init_pair(x_l, x_h)
    return _IWT.apply(x_l, x_h)


# ------------------------------------------------------------------------------------------------
# selective scan
# ------------------------------------------------------------------------------------------------
def _scan_shapes(u, delta, A, B, C, D, z, delta_bias):
    if u.dim() != 3:
        raise RuntimeError(f"selective_scan_fn: u must be (batch, dim, L), got {tuple(u.shape)}")
    batch, dim, L = u.shape
    if delta.shape != u.shape:
        raise RuntimeError(f"selective_scan_fn: delta shape {tuple(delta.shape)} != u shape {tuple(u.shape)}")
    if A.dim() != 2 or A.shape[0] != dim:
        raise RuntimeError(f"selective_scan_fn: A must be (dim, N) with dim={dim}, got {tuple(A.shape)}")
    N = A.shape[1]
    if B.dim() == 2 or C.dim() == 2:
        raise NotImplementedError("selective_scan_fn: time-invariant (dim, N) B/C are not implemented; "
                                  "the reference only uses (batch, G, N, L) B/C (wavemamba_arch.py:459-460)")
    if B.dim() == 3:
        B = B.unsqueeze(1)
    if C.dim() == 3:
        C = C.unsqueeze(1)
    G = B.shape[1]
    if B.shape != (batch, G, N, L) or C.shape != (batch, G, N, L):
        raise RuntimeError(f"selective_scan_fn: B/C must be (batch, G, N, L) = ({batch}, G, {N}, {L}), "
                           f"got {tuple(B.shape)} / {tuple(C.shape)}")
    if dim % G:
        raise RuntimeError(f"selective_scan_fn: dim {dim} not divisible by the number of groups {G}")
    for name, t in (("D", D), ("delta_bias", delta_bias)):
        if t is not None and t.shape != (dim,):
            raise RuntimeError(f"selective_scan_fn: {name} must be ({dim},), got {tuple(t.shape)}")
    if z is not None and z.shape != u.shape:
        raise RuntimeError("selective_scan_fn: z shape must match u")
    return batch, dim, L, N, G, B, C


def _f32c(t):
    return None if t is None else t.contiguous().float()


def _scan_forward(u, delta, A, B, C, D, z, delta_bias, delta_softplus, want_last):
    lib = _lib.load()
    batch, dim, L, N, G = u.shape[0], u.shape[1], u.shape[2], A.shape[1], B.shape[1]
    out = torch.empty_like(u)
    last = torch.empty((batch, dim, N), dtype=torch.float32, device=u.device) if want_last else None
    ws_bytes = lib.wm_selscan_fwd_workspace_bytes(batch, dim, L, N, G)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=u.device) if ws_bytes else None
    with torch.cuda.device(u.device):
        check(lib.wm_selscan_fwd(_ptr(u), _ptr(delta), _ptr(A), _ptr(B), _ptr(C), _ptr(D), _ptr(z),
                                 _ptr(delta_bias), _ptr(out), _ptr(last), _ptr(ws), ws_bytes,
                                 batch, dim, L, N, G, int(bool(delta_softplus)), _stream()),
              "wm_selscan_fwd")
    return out, last


class _SelectiveScan(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, delta, A, B, C, D, delta_bias, delta_softplus, want_last):
        out, last = _scan_forward(u, delta, A, B, C, D, None, delta_bias, delta_softplus, want_last)
        ctx.save_for_backward(u, delta, A, B, C, D, delta_bias)
        ctx.delta_softplus = bool(delta_softplus)
        if want_last:
            ctx.mark_non_differentiable(last)
            return out, last
        return out

    @staticmethod
    def backward(ctx, dout, *unused):
        lib = _lib.load()
        u, delta, A, B, C, D, delta_bias = ctx.saved_tensors
        batch, dim, L = u.shape
        N, G = A.shape[1], B.shape[1]
        dout = dout.contiguous().float()
        du = torch.empty_like(u)
        ddelta = torch.empty_like(delta)
        dA = torch.empty_like(A)
        dB = torch.empty_like(B)
        dC = torch.empty_like(C)
        dD = torch.empty_like(D) if D is not None else None
        dbias = torch.empty_like(delta_bias) if delta_bias is not None else None
        ws_bytes = lib.wm_selscan_bwd_workspace_bytes(batch, dim, L, N, G)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=u.device) if ws_bytes else None
        with torch.cuda.device(u.device):
            check(lib.wm_selscan_bwd(_ptr(u), _ptr(delta), _ptr(A), _ptr(B), _ptr(C), _ptr(D),
                                     _ptr(delta_bias), _ptr(dout), _ptr(du), _ptr(ddelta), _ptr(dA),
                                     _ptr(dB), _ptr(dC), _ptr(dD), _ptr(dbias), _ptr(ws), ws_bytes,
                                     batch, dim, L, N, G, int(ctx.delta_softplus), _stream()),
                  "wm_selscan_bwd")
        return du, ddelta, dA, dB, dC, dD, dbias, None, None


def selective_scan_fn(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                      return_last_state=False):
    """Drop-in for mamba_ssm.ops.selective_scan_interface.selective_scan_fn (reference call site
    wavemamba_arch.py:465-471).  u, delta: (batch, dim, L); A: (dim, N); B, C: (batch, [G,] N, L);
    D, delta_bias: (dim,).  Computes in fp32, returns `out` in u's dtype (and the fp32 last state
    (batch, dim, N) when return_last_state).  Differentiable w.r.t. u, delta, A, B, C, D, delta_bias
    (and z, through eager gating).  CPU tensors: cpu_twin.selective_scan_fn (plain PyTorch, same signature)."""
    if _on_cpu(u, delta, A, B, C, D, z, delta_bias):
        from . import cpu_twin
        return cpu_twin.selective_scan_fn(u, delta, A, B, C, D, z, delta_bias, delta_softplus, return_last_state)
    _require_cuda("selective_scan_fn", u, delta, A, B, C, D, z, delta_bias)
    batch, dim, L, N, G, B4, C4 = _scan_shapes(u, delta, A, B, C, D, z, delta_bias)
    squeeze_B, squeeze_C = B.dim() == 3, C.dim() == 3
    in_dtype = u.dtype
    args = [_f32c(t) for t in (u, delta, A, B4, C4, D, delta_bias)]
    need_grad = torch.is_grad_enabled() and any(t is not None and t.requires_grad
                                                for t in (u, delta, A, B, C, D, z, delta_bias))
    if not need_grad:
        out, last = _scan_forward(args[0], args[1], args[2], args[3], args[4], args[5], _f32c(z),
                                  args[6], delta_softplus, return_last_state)
    else:
        res = _SelectiveScan.apply(*args, bool(delta_softplus), bool(return_last_state))
        out, last = res if return_last_state else (res, None)
        if z is not None:
            out = out * F.silu(z.float())
    out = out.to(in_dtype)
    return (out, last) if return_last_state else out


# ------------------------------------------------------------------------------------------------
# fused SS2D four-direction core (forward / inference)
# ------------------------------------------------------------------------------------------------
def ss2d_core_supported(d_inner, d_state, dt_rank, width=None, height=None):
    """Shapes the fused HIP core covers (else: direction glue + selective_scan_fn).  Any map size (widths that are not
    a multiple of 4 take the kernels' element-wise tile accesses); with `width` and `height`, maps beyond the kernels'
    32-bit element offsets (d_inner * H * W >= 2^31, wavemamba_hip.hip: core_plan) are refused here instead of at the
    call."""
    if d_inner > 64 or dt_rank > 4 or d_state > 32:
        return False
    if width is not None and height is not None and d_inner * width * height > 2 ** 31 - 1:
        return False
    return True


def ss2d_core_bwd_supported(d_inner, d_state, dt_rank):
    """Shapes wm_ss2d_core_bwd covers (training): d_state <= 32 (BASELINE config 5's block included)."""
    return d_inner <= 64 and d_state <= 32 and dt_rank <= 4


def _ss2d_core_shapes(x, x_proj_weight, dt_projs_weight, A_logs):
    B, D, H, W = x.shape
    K, C, D2 = x_proj_weight.shape
    R, N = dt_projs_weight.shape[2], A_logs.shape[1]
    if K != 4 or D2 != D or C != R + 2 * N or dt_projs_weight.shape != (4, D, R) or A_logs.shape[0] != 4 * D:
        raise RuntimeError("ss2d_core: inconsistent parameter shapes")
    if not ss2d_core_supported(D, N, R, W, H):
        raise NotImplementedError(f"ss2d_core: d_inner={D}, d_state={N}, dt_rank={R} outside the fused kernel's range")
    return B, D, H, W, N, R


def _host_wait_under_capture(event):
    """Block the host until `event` (recorded before the capture began) has completed.  event.synchronize() is one of the
    calls that invalidate a capture in the global capture mode (hipErrorStreamCaptureUnsupported, then
    hipErrorStreamCaptureInvalidated at the next launch - torch.cuda.graph's default mode): the library does it under a
    thread-local relaxed mode (wm_event_synchronize_relaxed)."""
    check(_lib.load().wm_event_synchronize_relaxed(event.cuda_event), "wm_event_synchronize_relaxed")


_CORE_PREP_CACHE = {}   # id(x_proj_weight) -> (weakrefs, data_ptrs, versions, prepared buffer, done event, stream)


def _ss2d_core_prepared(params):
    """The prepared copy (wm_ss2d_core_prep: bf16-split x_proj fragments, A * log2(e), per-channel constants) of the five
    SS2D parameters `params` = (x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds), kept per module while the
    parameters are unchanged (`_version`, storage) - inference runs the core 14 times per image with static parameters.
    Returns None (the call prepares for itself) for parameters that require grad (training updates them every step),
    under graph capture, and for temporaries."""
    import weakref
    if any((not isinstance(p, torch.nn.Parameter)) or p.requires_grad and torch.is_grad_enabled() for p in params):
        return None
    key = id(params[0])
    cur = torch.cuda.current_stream(params[0].device)
    sig = tuple((p.data_ptr(), p._version, p.dtype) for p in params)
    ent = _CORE_PREP_CACHE.get(key)
    if ent is not None and all(r() is p for r, p in zip(ent[0], params)) and ent[1] == sig:
        if ent[4] != cur.cuda_stream:
            if torch.cuda.is_current_stream_capturing():
                # an event recorded outside the capture cannot be waited for inside it: block the HOST until the producer
                # (a warm-up forward on another stream that was never joined) is done, then the buffer is simply there
                _host_wait_under_capture(ent[3])
            else:
                cur.wait_event(ent[3])
                ent[2].record_stream(cur)
        return ent[2]
    if torch.cuda.is_current_stream_capturing():
        return None                                  # cold cache under capture: the call prepares for itself, inside the graph
    lib = _lib.load()
    f = [p.detach().contiguous().float() for p in params]
    D, R, N = f[1].shape[1], f[1].shape[2], f[3].shape[1]
    buf = torch.empty(lib.wm_ss2d_core_prep_bytes(N), dtype=torch.uint8, device=f[0].device)
    with torch.cuda.device(f[0].device):
        check(lib.wm_ss2d_core_prep(*[_ptr(t) for t in f], _ptr(buf), D, N, R, _stream()), "wm_ss2d_core_prep")
    ev = torch.cuda.Event()
    ev.record(cur)
    if key not in _CORE_PREP_CACHE:                  # one finalizer per parameter object, not one per rebuild
        weakref.finalize(params[0], _CORE_PREP_CACHE.pop, key, None)
    _CORE_PREP_CACHE[key] = ([weakref.ref(p) for p in params], sig, buf, ev, cur.cuda_stream)
    return buf


def _ss2d_core_fwd(f, merged, prepared=None, separate=False):
    """f[0] = x (fp32 or bf16 planes: the outputs take the same storage type), f[1:] fp32 parameters; `prepared`: the
    buffer of _ss2d_core_prepared for these parameters, or None.  separate: four allocations instead of one (4, B, D, L)
    block (torch.library ops must not return tensors that share a storage)."""
    lib = _lib.load()
    x = f[0]
    B, D, H, W, N, R = _ss2d_core_shapes(x, f[1], f[2], f[4])
    L = H * W
    if merged:
        outs = [torch.empty((B, D, L), dtype=x.dtype, device=x.device)]
    elif separate:
        outs = [torch.empty((B, D, L), dtype=x.dtype, device=x.device) for _ in range(4)]
    else:       # one allocation, reference return order: a consumer can add the four with one base pointer + stride
        outs = list(torch.empty((4, B, D, L), dtype=x.dtype, device=x.device).unbind(0))
    ws_bytes = lib.wm_ss2d_core_fwd_workspace_bytes(B, D, H, W, N, R, int(merged))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    ptrs = [_ptr(o) for o in outs] + [None] * (4 - len(outs))
    with torch.cuda.device(x.device):
        check(lib.wm_ss2d_core_fwd(*[_ptr(t) for t in f], *ptrs, int(merged), _ptr(ws), ws_bytes,
                                   None if prepared is None else _ptr(prepared),
                                   B, D, H, W, N, R, _dtype_code(x, "ss2d_core"), _stream()), "wm_ss2d_core_fwd")
    return outs


class _SS2DCoreFn(torch.autograd.Function):
    """SS2D.forward_core with HIP forward (wm_ss2d_core_fwd) and HIP backward (wm_ss2d_core_bwd).  Nothing but x and
    the parameters is saved: the backward re-runs the projection and the chunked scan from them."""

    @staticmethod
    def forward(ctx, merged, x, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds):
        f = [t.detach().contiguous().float() for t in (x, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds)]
        ctx.save_for_backward(*f)
        ctx.merged = merged
        outs = _ss2d_core_fwd(f, merged)
        return outs[0] if merged else tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        lib = _lib.load()
        f = ctx.saved_tensors
        x = f[0]
        B, D, H, W, N, R = _ss2d_core_shapes(x, f[1], f[2], f[4])
        if ctx.merged:
            g = dys[0].contiguous().float()
            dy = [g, g, g, g]
        else:
            dy = [(torch.zeros((B, D, H * W), dtype=torch.float32, device=x.device) if g is None else g.contiguous().float())
                  for g in dys]
        outs = [torch.empty_like(x), torch.empty_like(f[1]), torch.empty_like(f[2]), torch.empty_like(f[3]),
                torch.empty_like(f[4]), torch.empty_like(f[5])]
        ws_bytes = lib.wm_ss2d_core_bwd_workspace_bytes(B, D, H, W, N, R)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            check(lib.wm_ss2d_core_bwd(*[_ptr(t) for t in f], *[_ptr(t) for t in dy], *[_ptr(t) for t in outs],
                                       _ptr(ws), ws_bytes, B, D, H, W, N, R, _stream()), "wm_ss2d_core_bwd")
        return (None, *outs)


def ss2d_core(x, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds, merged=False):
    """SS2D.forward_core (reference wavemamba_arch.py:446-478) in one call.
    x (B, D, H, W) fp32 -> (y_row_fwd, y_row_rev, y_col_fwd, y_col_rev), each (B, D, H*W) in
    row-major l - the reference's return order; merged=True returns their sum (what :490 computes).
    Differentiable w.r.t. x and the five parameters (HIP backward, wm_ss2d_core_bwd)."""
    _lib.load()
    _require_cuda("ss2d_core", x, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds)
    _ss2d_core_shapes(x, x_proj_weight, dt_projs_weight, A_logs)
    args = (x, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds)
    if torch.is_grad_enabled() and any(t.requires_grad for t in args):
        return _SS2DCoreFn.apply(bool(merged), *args)
    xin = x.detach().contiguous()
    if xin.dtype != torch.bfloat16:                # bf16 planes stay bf16 (outputs too); anything else computes in fp32
        xin = xin.float()
    outs = _ss2d_core_fwd([xin] + [t.detach().contiguous().float() for t in args[1:]], int(bool(merged)),
                          _ss2d_core_prepared(args[1:]))
    return outs[0] if merged else tuple(outs)


# ------------------------------------------------------------------------------------------------
# LFSSBlock forward, fused (inference): lfss_in -> dwconv+SiLU -> ss2d core -> lfss_mid -> dwconv -> lfss_out
# ------------------------------------------------------------------------------------------------
def lfss_block_supported(C, d_inner, d_state, dt_rank, ffn_hidden, width=None, height=None):
    return (C in (8, 16, 32) and d_inner == 2 * C and ffn_hidden == 2 * C
            and ss2d_core_supported(d_inner, d_state, dt_rank, width, height))


def _w(t):
    return t.detach().contiguous().float()


_FUSE_OUT_CONV = True      # tests / tools: False takes wm_dwconv3x3_fwd + wm_lfss_out_fwd (bit-identical on fp32 planes)
# (ln_1 -> in_proj -> depth-wise 3x3 -> SiLU as ONE kernel was built in round 4 - wm_lfss_in_conv_fwd, parity-green - and measured
# 44 % slower than the two streaming kernels, profiles/r04/lfss_prologue_one_kernel.txt; deleted in round 5.)
# The gate z = in_proj(ln_1(x))[D:] recomputed by lfss_mid from the tokens instead of written by lfss_in and read back
# (512 of the block's 3456 B per position; bit-identical in fp32 planes: tests/test_gpu_parity.py).  0: round-3 data flow.
_RECOMPUTE_Z = True


def lfss_block_forward(tok, x_size, blk, tok_nchw=False, out_nchw=False):
    """LFSSBlock.forward (reference wavemamba_arch.py:520-528) on the HIP path, forward only.
    tok: (B, L, C) tokens, or (B, C, H, W) when tok_nchw.  `blk` supplies the parameters (an LFSSBlock
    module: ln_1, self_attention, skip_scale, conv_blk, ln_2, skip_scale2).  Returns (B, L, C) tokens or
    (B, C, H, W) when out_nchw."""
    lib = _lib.load()
    _require_cuda("lfss_block_forward", tok)
    H, W = x_size
    L = H * W
    ss, ff = blk.self_attention, blk.conv_blk
    C, D = ss.d_model, ss.d_inner
    B = tok.shape[0]
    tok = tok.contiguous().float()
    dev = tok.device
    st = _stream()
    # bf16 planes: the C = 32 kernels on maps with 16-byte tile accesses (W % 4 == 0); else fp32 planes
    pd = _PLANE_DTYPE if (C == 32 and W % 4 == 0) else torch.float32
    code = WM_F32 if pd == torch.float32 else WM_BF16
    # C == 32: the gate z is recomputed by the block's middle kernel from the tokens it reads anyway (wm_lfss_mid_rz_fwd, bit-identical
    # in fp32 planes) - lfss_in writes the x half only (ops._RECOMPUTE_Z = False: the written / re-read z, tests and tools)
    rz = C == 32 and _RECOMPUTE_Z
    z = None if rz else torch.empty((B, D, L), dtype=pd, device=dev)
    x = torch.empty((B, D, H, W), dtype=pd, device=dev)
    with torch.cuda.device(dev):
        check(lib.wm_lfss_in_fwd(_ptr(tok), int(tok_nchw), _ptr(_w(blk.ln_1.weight)), _ptr(_w(blk.ln_1.bias)),
                                 float(blk.ln_1.eps), _ptr(_w(ss.in_proj.weight)), _ptr(x), _ptr(z), B, L, C, code, st),
              "wm_lfss_in_fwd")
    xc = dwconv3x3(x, ss.conv2d.weight, ss.conv2d.bias, "silu")
    # the four directions' outputs stay separate (one (4, B, D, L) allocation); lfss_mid adds them as it loads (:490)
    core_params = (ss.x_proj_weight, ss.dt_projs_weight, ss.dt_projs_bias, ss.A_logs, ss.Ds)
    ny = 4
    y4 = _ss2d_core_fwd([xc] + [_w(t) for t in core_params], merged=0, prepared=_ss2d_core_prepared(core_params))
    tok1 = torch.empty((B, L, C), dtype=torch.float32, device=dev)
    f = torch.empty((B, D, H, W), dtype=pd, device=dev)
    with torch.cuda.device(dev):
        if rz:
            check(lib.wm_lfss_mid_rz_fwd(_ptr(y4[0]), ny, B * D * L, _ptr(tok), int(tok_nchw), _ptr(_w(blk.ln_1.weight)),
                                         _ptr(_w(blk.ln_1.bias)), float(blk.ln_1.eps), _ptr(_w(ss.in_proj.weight)),
                                         _ptr(_w(ss.out_norm.weight)), _ptr(_w(ss.out_norm.bias)), float(ss.out_norm.eps),
                                         _ptr(_w(ss.out_proj.weight)), _ptr(_w(blk.skip_scale)), _ptr(_w(blk.ln_2.weight)),
                                         _ptr(_w(blk.ln_2.bias)), float(blk.ln_2.eps), _ptr(_w(ff.conv1.weight)),
                                         _ptr(_w(ff.conv1.bias)), _ptr(tok1), _ptr(f), B, L, C, code, st), "wm_lfss_mid_rz_fwd")
        else:
            check(lib.wm_lfss_mid_fwd(_ptr(y4[0]), ny, B * D * L, _ptr(z), _ptr(tok), int(tok_nchw), _ptr(_w(ss.out_norm.weight)),
                                      _ptr(_w(ss.out_norm.bias)), float(ss.out_norm.eps), _ptr(_w(ss.out_proj.weight)),
                                      _ptr(_w(blk.skip_scale)), _ptr(_w(blk.ln_2.weight)), _ptr(_w(blk.ln_2.bias)),
                                      float(blk.ln_2.eps), _ptr(_w(ff.conv1.weight)), _ptr(_w(ff.conv1.bias)),
                                      _ptr(tok1), _ptr(f), B, L, C, code, st), "wm_lfss_mid_fwd")
    out = torch.empty((B, C, H, W) if out_nchw else (B, L, C), dtype=torch.float32, device=dev)
    if C == 32 and W % 32 == 0 and _FUSE_OUT_CONV:
        # the ffn's depth-wise 3x3 inside the closing kernel: fc (conv2's output) never reaches HBM
        with torch.cuda.device(dev):
            check(lib.wm_lfss_out_conv_fwd(_ptr(f), _ptr(_w(ff.conv2.weight)),
                                           None if ff.conv2.bias is None else _ptr(_w(ff.conv2.bias)), _ptr(tok1),
                                           _ptr(_w(ff.conv3.weight)), _ptr(_w(ff.conv3.bias)), _ptr(_w(blk.skip_scale2)),
                                           _ptr(out), int(out_nchw), B, H, W, C, code, st), "wm_lfss_out_conv_fwd")
        return out
    fc = dwconv3x3(f, ff.conv2.weight, ff.conv2.bias, "none")
    with torch.cuda.device(dev):
        check(lib.wm_lfss_out_fwd(_ptr(fc), _ptr(tok1), _ptr(_w(ff.conv3.weight)), _ptr(_w(ff.conv3.bias)),
                                  _ptr(_w(blk.skip_scale2)), _ptr(out), int(out_nchw), B, L, C, code, st),
              "wm_lfss_out_fwd")
    return out


# ------------------------------------------------------------------------------------------------
# LayerNorm2d (HFE branch, forward only)
# ------------------------------------------------------------------------------------------------
def layernorm2d(x, weight, bias, eps):
    """Per-pixel LayerNorm over channels of an NCHW fp32 map (reference LayerNorm2d, :532-569)."""
    lib = _lib.load()
    _require_cuda("layernorm2d", x, weight, bias)
    B, C, H, W = x.shape
    if C not in (8, 16, 32, 64) or x.dtype != torch.float32:
        raise NotImplementedError("layernorm2d: fp32, C in {8, 16, 32, 64}")
    x = x.contiguous()
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        check(lib.wm_layernorm2d_fwd(_ptr(x), _ptr(_w(weight)), _ptr(_w(bias)), float(eps), _ptr(y), B, H * W, C,
                                     _stream()), "wm_layernorm2d_fwd")
    return y


# ------------------------------------------------------------------------------------------------
# channel Gram matrix over pixels (HFE branch, forward only)
# ------------------------------------------------------------------------------------------------
def gram(x, y):
    """x, y (B, C, L) fp32, C <= 32 -> (G (B, C, C) = x @ y^T over L, |x_i|^2 (B, C), |y_j|^2 (B, C))."""
    lib = _lib.load()
    _require_cuda("gram", x, y)
    B, C, L = x.shape
    if y.shape != x.shape or C > 32 or x.dtype != torch.float32 or y.dtype != torch.float32:
        raise NotImplementedError("gram: two fp32 (B, C<=32, L) tensors of equal shape")
    x, y = x.contiguous(), y.contiguous()
    buf = torch.empty(B * C * (C + 2), dtype=torch.float32, device=x.device)
    G = buf[:B * C * C].view(B, C, C)
    nx = buf[B * C * C:B * C * (C + 1)].view(B, C)
    ny = buf[B * C * (C + 1):].view(B, C)
    nws = int(lib.wm_gram_workspace_bytes(B, C, L))
    ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        check(lib.wm_gram_fwd(_ptr(x), _ptr(y), _ptr(G), _ptr(nx), _ptr(ny), _ptr(ws), nws, B, C, L, _stream()),
              "wm_gram_fwd")
    return G, nx, ny


class _GramTrain(torch.autograd.Function):
    """gram() under autograd: (G, |x|^2, |y|^2) of x, y (B, C, L).  Backward: dx = dG y + 2 d|x|^2 x, dy = dG^T x + 2 d|y|^2 y
    (two batched GEMMs and two scaled adds).  The training-side form of the transposed attention's
    normalize(q) @ normalize(k)^T (reference :783-786): one pass over q and k instead of two normalisations and a batched
    product, and the division by the norms happens on the (B, C, C) result."""

    @staticmethod
    def forward(ctx, x, y):
        x, y = x.contiguous(), y.contiguous()
        ctx.save_for_backward(x, y)
        G, nx, ny = gram(x, y)
        return G, nx, ny

    @staticmethod
    def backward(ctx, dG, dnx, dny):
        x, y = ctx.saved_tensors
        dx = dy = None
        if ctx.needs_input_grad[0]:
            dx = torch.bmm(dG, y) if dG is not None else torch.zeros_like(x)
            if dnx is not None:
                dx.addcmul_(x, dnx.unsqueeze(2), value=2.0)
        if ctx.needs_input_grad[1]:
            dy = torch.bmm(dG.transpose(1, 2), x) if dG is not None else torch.zeros_like(y)
            if dny is not None:
                dy.addcmul_(y, dny.unsqueeze(2), value=2.0)
        return dx, dy


def gram_train(x, y):
    """gram(x, y) with gradients (x, y (B, C <= 32, L) fp32 HIP tensors)."""
    return _GramTrain.apply(x, y)


class _DWConvTrain(torch.autograd.Function):
    """Depth-wise 3x3 conv (+bias) with HIP forward, input gradient (the same kernel on the flipped weight) and
    weight / bias gradient (strip reduction).  Training-side replacement of MIOpen's naive depth-wise kernels."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return dwconv3x3(x, weight, bias, "none")

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x, weight = ctx.saved_tensors
        gy = gy.contiguous().float()
        B, C, H, W = x.shape
        gx = dwconv3x3(gy, weight, None, "none", flip=True)
        buf = _zeros_small(10 * C, x.device)                                    # dW | db back to back, zeroed
        dW = buf[:9 * C].view(weight.shape)
        db = buf[9 * C:] if ctx.has_bias else None
        with torch.cuda.device(x.device):
            check(lib.wm_dwconv3x3_wgrad(_ptr(x.contiguous()), _ptr(gy), _ptr(dW), _ptr(db), B, C, H, W, _stream()),
                  "wm_dwconv3x3_wgrad")
        return gx, dW, db


def dwconv3x3_train(x, weight, bias=None):
    """Differentiable depth-wise 3x3 conv (stride 1, padding 1) on the HIP path (fp32 NCHW)."""
    _require_cuda("dwconv3x3_train", x, weight, bias)
    return _DWConvTrain.apply(x.contiguous().float(), weight, bias)


class _LayerNorm2dTrain(torch.autograd.Function):
    """LayerNorm2d with HIP forward and backward (reference LayerNormFunction, wavemamba_arch.py:532-557)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        ctx.save_for_backward(x, weight)
        ctx.eps = float(eps)
        return layernorm2d(x, weight, bias, eps)

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x, weight = ctx.saved_tensors
        B, C, H, W = x.shape
        gy = gy.contiguous().float()
        gx = torch.empty_like(x)
        buf = _zeros_small(2 * C, x.device)                                     # dweight | dbias back to back, zeroed
        dw, db = buf[:C], buf[C:]
        with torch.cuda.device(x.device):
            check(lib.wm_layernorm2d_bwd(_ptr(x), _ptr(_w(weight)), _ptr(gy), ctx.eps, _ptr(gx), _ptr(dw), _ptr(db),
                                         B, H * W, C, _stream()), "wm_layernorm2d_bwd")
        return gx, dw, db, None


def layernorm2d_train(x, weight, bias, eps):
    _require_cuda("layernorm2d_train", x, weight, bias)
    return _LayerNorm2dTrain.apply(x.contiguous().float(), weight, bias, eps)


class _LayerNormTok(torch.autograd.Function):
    """nn.LayerNorm(C) over the last axis of a contiguous (..., C) fp32 tensor: HIP forward and backward."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        lib = _lib.load()
        C = x.shape[-1]
        w, b = weight.detach().contiguous().float(), bias.detach().contiguous().float()
        y = torch.empty_like(x)
        with torch.cuda.device(x.device):
            check(lib.wm_layernorm_tok_fwd(_ptr(x), _ptr(w), _ptr(b), float(eps), _ptr(y), x.numel() // C, C, _stream()),
                  "wm_layernorm_tok_fwd")
        ctx.save_for_backward(x, w)
        ctx.eps = float(eps)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        C = x.shape[-1]
        gy = gy.contiguous().float()
        gx = torch.empty_like(x)
        buf = _zeros_small(2 * C, x.device)
        dw, db = buf[:C], buf[C:]
        with torch.cuda.device(x.device):
            check(lib.wm_layernorm_tok_bwd(_ptr(x), _ptr(w), _ptr(gy), ctx.eps, _ptr(gx), _ptr(dw), _ptr(db),
                                           x.numel() // C, C, _stream()), "wm_layernorm_tok_bwd")
        return gx, dw, db, None


def layernorm_tok_supported(x, C):
    return x.is_cuda and x.dtype == torch.float32 and x.shape[-1] == C and C in (8, 16, 32, 64)


def layernorm_tok(x, weight, bias, eps):
    """F.layer_norm(x, (C,), weight, bias, eps) for a (..., C) fp32 tensor, C in {8, 16, 32, 64}; differentiable
    (HIP backward: input gradient + weight / bias gradients in one pass)."""
    _require_cuda("layernorm_tok", x, weight, bias)
    x = x.contiguous().float()
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or bias.requires_grad):
        return _LayerNormTok.apply(x, weight, bias, eps)
    return _LayerNormTok.forward(_NoCtx(), x, weight, bias, eps)


class _NoCtx:
    def save_for_backward(self, *a):
        pass


# ------------------------------------------------------------------------------------------------
# element-wise gates and scaled skips of the LFSSBlock training path (forward + backward in HIP)
# ------------------------------------------------------------------------------------------------
_ACTS = {"silu": 1, "gelu": 2, "sigmoid": 3}


def _batch_strided(t):
    """(B, ...) tensor whose batch items are dense: -> batch stride in elements, or None."""
    if t.dim() < 2 or t.dtype != torch.float32:
        return None
    inner = t[0]
    return t.stride(0) if inner.is_contiguous() else None


class _Gate(torch.autograd.Function):
    """out = act(a) * b (wm_gate_fwd / wm_gate_bwd): SS2D's `y * F.silu(z)` (reference :493) and the ffn's
    `F.gelu(x1) * x2` (:228-229).  a, b: (B, C, H, W) fp32, each dense per batch item (channel-chunk views are fine)."""

    @staticmethod
    def forward(ctx, a, b, act):
        lib = _lib.load()
        B, per_b = a.shape[0], a[0].numel()
        out = torch.empty(a.shape, dtype=torch.float32, device=a.device)
        with torch.cuda.device(a.device):
            check(lib.wm_gate_fwd(_ptr(a), _ptr(b), _ptr(out), act, B, per_b, a.stride(0), b.stride(0), out.stride(0) if B else 0,
                                  _stream()), "wm_gate_fwd")
        ctx.save_for_backward(a, b)
        ctx.act = act
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        a, b = ctx.saved_tensors
        g = g.contiguous().float()
        B, per_b = a.shape[0], a[0].numel()
        ga, gb = torch.empty_like(g), torch.empty_like(g)
        with torch.cuda.device(a.device):
            check(lib.wm_gate_bwd(_ptr(a), _ptr(b), _ptr(g), _ptr(ga), _ptr(gb), ctx.act, B, per_b, a.stride(0), b.stride(0),
                                  g.stride(0) if B else 0, ga.stride(0) if B else 0, gb.stride(0) if B else 0, _stream()),
                  "wm_gate_bwd")
        return ga, gb, None


class _GluGate(torch.autograd.Function):
    """out = act(t[:, :C]) * t[:, C:] for t (B, 2C, H, W): the gated ffn's chunk + gate as one node, so that the backward
    writes ONE gradient tensor in t's layout (autograd's backward of a chunk would concatenate two)."""

    @staticmethod
    def forward(ctx, t, act):
        lib = _lib.load()
        t = t.contiguous().float()
        B, C2 = t.shape[:2]
        C = C2 // 2
        per_b = C * t[0, 0].numel()
        out = torch.empty((B, C) + tuple(t.shape[2:]), dtype=torch.float32, device=t.device)
        with torch.cuda.device(t.device):
            check(lib.wm_gate_fwd(_ptr(t), t.data_ptr() + 4 * per_b, _ptr(out), act, B, per_b, 2 * per_b, 2 * per_b, per_b,
                                  _stream()), "wm_gate_fwd")
        ctx.save_for_backward(t)
        ctx.act = act
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (t,) = ctx.saved_tensors
        g = g.contiguous().float()
        B, per_b = t.shape[0], g[0].numel()
        gt = torch.empty_like(t)
        with torch.cuda.device(t.device):
            check(lib.wm_gate_bwd(_ptr(t), t.data_ptr() + 4 * per_b, _ptr(g), _ptr(gt), gt.data_ptr() + 4 * per_b, ctx.act, B,
                                  per_b, 2 * per_b, 2 * per_b, per_b, 2 * per_b, 2 * per_b, _stream()), "wm_gate_bwd")
        return gt, None


def glu_gate(t, act):
    """act(t[:, :C]) * t[:, C:] for a (B, 2C, H, W) fp32 tensor, act in {"silu", "gelu"}, differentiable (_GluGate)."""
    _lib.load()
    _require_cuda("glu_gate", t)
    if t.dim() != 4 or t.shape[1] % 2 or t.shape[0] > 65535:
        raise NotImplementedError("glu_gate: a (B, 2C, H, W) tensor")
    return _GluGate.apply(t, _ACTS[act])


def gate_supported(a, b):
    return (a.is_cuda and b.is_cuda and a.shape == b.shape and a.dim() == 4 and _batch_strided(a) is not None
            and _batch_strided(b) is not None and a.shape[0] <= 65535)


def gate_act(a, b, act):
    """act(a) * b with act in {"silu", "gelu", "sigmoid"} (exact erf GELU), differentiable; see _Gate."""
    _lib.load()
    _require_cuda("gate_act", a, b)
    if not gate_supported(a, b):
        raise NotImplementedError("gate_act: two fp32 (B, C, H, W) tensors, dense per batch item")
    return _Gate.apply(a, b, _ACTS[act])


class _ScaleAdd(torch.autograd.Function):
    """out = x * scale[c] + o over (B, C, H, W) (wm_scale_add_fwd / _bwd): LFSSBlock's scaled skips (reference :525-526)."""

    @staticmethod
    def forward(ctx, x, scale, o):
        lib = _lib.load()
        x, o, sc = x.contiguous().float(), o.contiguous().float(), scale.detach().contiguous().float()
        B, C = x.shape[:2]
        L = x[0, 0].numel()
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            check(lib.wm_scale_add_fwd(_ptr(x), _ptr(sc), _ptr(o), _ptr(out), B, C, L, _stream()), "wm_scale_add_fwd")
        ctx.save_for_backward(x, sc)
        ctx.scale_shape = scale.shape
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, sc = ctx.saved_tensors
        g = g.contiguous().float()
        B, C = x.shape[:2]
        L = x[0, 0].numel()
        gx, gs = torch.empty_like(x), _zeros_small(C, x.device)
        with torch.cuda.device(x.device):
            check(lib.wm_scale_add_bwd(_ptr(g), _ptr(x), _ptr(sc), _ptr(gx), _ptr(gs), B, C, L, _stream()), "wm_scale_add_bwd")
        return gx, gs.view(ctx.scale_shape), g


def scale_add(x, scale, o):
    """x * scale.view(1, C, 1, 1) + o for (B, C, H, W) fp32 tensors and a C-element scale, differentiable."""
    _lib.load()
    _require_cuda("scale_add", x, scale, o)
    if x.dim() != 4 or x.shape != o.shape or scale.numel() != x.shape[1] or x.shape[0] * x.shape[1] > 65535:
        raise NotImplementedError("scale_add: (B, C, H, W) operands with B C <= 65535 and a C-element scale")
    return _ScaleAdd.apply(x, scale, o)


# ------------------------------------------------------------------------------------------------
# depth-wise 3x3 convolution (+ bias, + SiLU) - inference path of SS2D.conv2d / ffn.conv2
# ------------------------------------------------------------------------------------------------
def dwconv3x3(x, weight, bias=None, act="none", flip=False):
    """F.conv2d(x, weight, bias, stride=1, padding=1, groups=C) [+ SiLU / exact GELU when act == 'silu' / 'gelu'] for a
    (C, 1, 3, 3) weight, NCHW fp32 (or bf16 planes: fp32 arithmetic, bf16 storage), forward only (no autograd graph is
    recorded).  flip: with weight.flip(2, 3) - the convolution's input gradient - read from `weight` itself."""
    lib = _lib.load()
    _require_cuda("dwconv3x3", x, weight, bias)
    B, C, H, W = x.shape
    if weight.shape != (C, 1, 3, 3):
        raise RuntimeError(f"dwconv3x3: weight must be ({C}, 1, 3, 3), got {tuple(weight.shape)}")
    code = _dtype_code(x, "dwconv3x3")
    x = x.contiguous()
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        check(lib.wm_dwconv3x3_fwd(_ptr(x), _ptr(weight.detach().contiguous().float()),
                                   _ptr(None if bias is None else bias.detach().contiguous().float()), _ptr(y),
                                   B, C, H, W, {"none": 0, "silu": 1, "gelu": 2}[act] + (4 if flip else 0), code, _stream()),
              "wm_dwconv3x3_fwd")
    return y


def image_pre_u8(img, window_size=128, swap_rb=True):
    """(h, w, 3) uint8 device image -> (1, 3, Hp, Wp) fp32 in [0, 1], channel-major, reflect-padded to multiples of
    `window_size` (inference_wavemamba.py:99-105 + :28-36); swap_rb: BGR (cv2) -> RGB."""
    lib = _lib.load()
    _require_cuda("image_pre_u8", img)
    if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3:
        raise RuntimeError(f"image_pre_u8: expected (h, w, 3) uint8, got {tuple(img.shape)} {img.dtype}")
    h, w = img.shape[:2]
    Hp, Wp = h + (window_size - h % window_size) % window_size, w + (window_size - w % window_size) % window_size
    out = torch.empty((1, 3, Hp, Wp), dtype=torch.float32, device=img.device)
    with torch.cuda.device(img.device):
        check(lib.wm_image_pre_u8(_ptr(img.contiguous()), _ptr(out), h, w, Hp, Wp, int(bool(swap_rb)), _stream()),
              "wm_image_pre_u8")
    return out


def image_post_u8(t, h, w, swap_rb=True):
    """(1, 3, Hp, Wp) fp32 -> (h, w, 3) uint8: crop, clamp [0, 1], * 255, round half to even, channel-last
    (inference_wavemamba.py:112-113 + tensor2img, img_util.py:67-94); swap_rb: RGB -> BGR."""
    lib = _lib.load()
    _require_cuda("image_post_u8", t)
    if t.dtype != torch.float32 or t.dim() != 4 or t.shape[0] != 1 or t.shape[1] != 3:
        raise RuntimeError(f"image_post_u8: expected (1, 3, Hp, Wp) float32, got {tuple(t.shape)} {t.dtype}")
    Hp, Wp = t.shape[2:]
    out = torch.empty((h, w, 3), dtype=torch.uint8, device=t.device)
    with torch.cuda.device(t.device):
        check(lib.wm_image_post_u8(_ptr(t.contiguous()), _ptr(out), h, w, Hp, Wp, int(bool(swap_rb)), _stream()),
              "wm_image_post_u8")
    return out


def match_index(G, nx, ny):
    """Channel matching with every channel kept: (B, C) int32 index of the L2-nearest candidate channel from the
    Gram outputs of `gram(maps, candidates)` (argmin_j |x_c|^2 + |y_j|^2 - 2 x_c . y_j)."""
    lib = _lib.load()
    _require_cuda("match_index", G, nx, ny)
    B, C = nx.shape
    idx = torch.empty((B, C), dtype=torch.int32, device=G.device)
    with torch.cuda.device(G.device):
        check(lib.wm_match_index(_ptr(G.contiguous()), _ptr(nx.contiguous()), _ptr(ny.contiguous()), _ptr(idx), B, C,
                                 _stream()), "wm_match_index")
    return idx


def attn_fold(G, nq, nk, temperature, w_po, batch, heads):
    """(batch, C, C) = w_po @ blockdiag_h softmax(G / (|q||k|) * temperature): the transposed attention folded into its
    1x1 output projection.  G (batch * heads, ch, ch), nq / nk (batch * heads, ch) squared norms."""
    lib = _lib.load()
    _require_cuda("attn_fold", G, nq, nk, temperature, w_po)
    C = w_po.shape[0]
    out = torch.empty((batch, C, C), dtype=torch.float32, device=G.device)
    with torch.cuda.device(G.device):
        check(lib.wm_attn_fold(_ptr(G.contiguous()), _ptr(nq.contiguous()), _ptr(nk.contiguous()),
                               _ptr(temperature.detach().contiguous()), _ptr(w_po.detach().reshape(C, C).contiguous()),
                               _ptr(out), batch, C, heads, _stream()), "wm_attn_fold")
    return out


def skff(x0, x1, x2, w_du, prelu, w_fc):
    """SKFF of three (B, C, H, W) sub-bands: w_du (d, C[, 1, 1]), prelu (1,), w_fc (3, C, d) -> (B, C, H, W)."""
    lib = _lib.load()
    _require_cuda("skff", x0, x1, x2, w_du, prelu, w_fc)
    B, C, H, W = x0.shape
    d = w_du.shape[0]
    if x1.shape != x0.shape or x2.shape != x0.shape or x0.dtype != torch.float32:
        raise RuntimeError("skff: three fp32 tensors of equal shape")
    x0, x1, x2 = x0.contiguous(), x1.contiguous(), x2.contiguous()
    out = torch.empty_like(x0)
    ws = torch.empty(lib.wm_skff_workspace_bytes(B, C), dtype=torch.uint8, device=x0.device)
    with torch.cuda.device(x0.device):
        check(lib.wm_skff_fwd(_ptr(x0), _ptr(x1), _ptr(x2), _ptr(w_du.detach().reshape(d, C).contiguous()),
                              _ptr(prelu.detach().contiguous()), _ptr(w_fc.detach().reshape(3, C, d).contiguous()),
                              _ptr(out), _ptr(ws), ws.numel(), B, C, d, H, W, _stream()), "wm_skff_fwd")
    return out


_WFRAG_CACHE = {}      # id(weight) -> (weakref, data_ptr, version, wfrag tensor, prep-done event, stream it was built on)


CONV3X3_AUTO, CONV3X3_FIRST_GEN, CONV3X3_WAVE_SPECIALISED = 0, 1, 2


def conv2d_select(mode=CONV3X3_AUTO):
    """Which 3x3 kernel serves `conv2d` / `conv2d_gated` (wm_conv2d_select): by problem size (default), always the
    first-generation kernel, or the persistent wave-specialised one wherever its limits allow.  Process-wide; the two
    kernels accumulate in the same order, so this never changes a result (tests assert bit-equality)."""
    check(_lib.load().wm_conv2d_select(int(mode)), "wm_conv2d_select")


def conv2d_cache_clear():
    """Drop every prepared (bf16-split) weight copy.  The cache validates an entry by (object, data_ptr, _version); a
    write through `.data` (p.data.copy_(), basicsr-style EMA `.data.mul_().add_()`, weight surgery) does NOT bump the
    version counter, so code that updates weights that way must call this (WaveMamba.train() / .eval() / ._apply() and
    trainer.load_network do)."""
    _WFRAG_CACHE.clear()
    _CORE_PREP_CACHE.clear()                 # (the SS2D core's prepared parameters: same validation, same caveat)


def _conv2d_wfrag(weight, cache=True):
    """The prepared (bf16-split, fragment-ordered) copy of a (Cout, Cin, ks, ks) weight; rebuilt when the
    parameter is updated in place (`_version`) or re-allocated.  cache=False: a weight computed on the fly
    (the folded attention), prepared every time.  A cached copy built on another stream is waited for (event) and
    recorded on the using stream, so multi-stream serving never reads it before the preparation kernel has finished
    nor sees it freed under a pending launch."""
    import weakref
    key = id(weight)
    cur = torch.cuda.current_stream(weight.device)
    ent = _WFRAG_CACHE.get(key) if cache else None
    if ent is not None and ent[0]() is weight and ent[1] == weight.data_ptr() and ent[2] == weight._version:
        if ent[5] != cur.cuda_stream:
            if torch.cuda.is_current_stream_capturing():
                # an event recorded outside the capture cannot be waited for inside it: block the HOST until the producer
                # (a warm-up forward on another stream that was never joined) is done - then the fragments are simply there
                _host_wait_under_capture(ent[4])
            else:
                cur.wait_event(ent[4])
                ent[3].record_stream(cur)
        return ent[3]
    lib = _lib.load()
    cout, cin, ks = weight.shape[0], weight.shape[1], weight.shape[2]
    frag = torch.empty(lib.wm_conv2d_wfrag_bytes(cout, cin, ks), dtype=torch.uint8, device=weight.device)
    with torch.cuda.device(weight.device):
        check(lib.wm_conv2d_prep(_ptr(weight.detach().contiguous()), _ptr(frag), cout, cin, ks, _stream()),
              "wm_conv2d_prep")
    if cache and torch.cuda.is_current_stream_capturing():
        return frag                                   # prepared inside the capture: part of the graph, not of the cache
    if cache:
        ev = torch.cuda.Event()
        ev.record(cur)
        _WFRAG_CACHE[key] = (weakref.ref(weight, lambda _r, k=key: _WFRAG_CACHE.pop(k, None)), weight.data_ptr(),
                             weight._version, frag, ev, cur.cuda_stream)
    return frag


def conv2d(x, weight, bias=None, x2=None, x2_index=None, gate=None, residual=None, dynamic_weight=False):
    """y = F.conv2d(X, weight, bias, stride=1, padding=ks // 2) for a dense (Cout, Cin, ks, ks) weight, ks in
    {1, 3}, NCHW fp32, where X = x, or cat([x, x2], 1), or cat([x, gather(x2, 1, x2_index)], 1) with x2_index
    (B, Cb) channel indices into x2; then y *= sigmoid(gate) and y += residual when given.  Forward only (no
    autograd graph is recorded).  bf16 matrix cores with a two-term split of both operands: 3-4e-6 relative to
    the fp64 result.  dynamic_weight: `weight` is a freshly computed tensor (no prepared-copy cache)."""
    lib = _lib.load()
    _require_cuda("conv2d", x, weight, bias, x2, x2_index, gate, residual)
    B, Ca, H, W = x.shape
    cout, cin, ks = weight.shape[0], weight.shape[1], weight.shape[2]
    if weight.dim() != 4 or weight.shape[3] != ks or ks not in (1, 3):
        raise RuntimeError(f"conv2d: weight must be (Cout, Cin, ks, ks) with ks in (1, 3), got {tuple(weight.shape)}")
    cb = cb_src = 0
    if x2 is not None:
        if (x2.shape[0], x2.shape[2], x2.shape[3]) != (B, H, W):
            raise RuntimeError(f"conv2d: x2 {tuple(x2.shape)} does not match x {tuple(x.shape)}")
        cb_src = x2.shape[1]
        cb = cb_src if x2_index is None else x2_index.shape[1]
        if x2_index is not None and (x2_index.shape[0] != B or x2_index.dim() != 2):
            raise RuntimeError(f"conv2d: x2_index must be (B, Cb), got {tuple(x2_index.shape)}")
    if cin != Ca + cb:
        raise RuntimeError(f"conv2d: weight expects {cin} input channels, got {Ca} + {cb}")
    for t in (x, weight, x2, gate, residual):
        if t is not None and t.dtype != torch.float32:
            raise RuntimeError("conv2d: float32 only")
    for t, name in ((gate, "gate"), (residual, "residual")):
        if t is not None and tuple(t.shape) != (B, cout, H, W):
            raise RuntimeError(f"conv2d: {name} must be {(B, cout, H, W)}, got {tuple(t.shape)}")
    x = x.contiguous()
    x2 = None if x2 is None else x2.contiguous()
    idx = None if x2_index is None else x2_index.to(torch.int32).contiguous()
    gate = None if gate is None else gate.contiguous()
    residual = None if residual is None else residual.contiguous()
    frag = _conv2d_wfrag(weight, cache=not dynamic_weight)
    y = torch.empty((B, cout, H, W), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(lib.wm_conv2d_fwd(_ptr(x), _ptr(x2), _ptr(idx), _ptr(frag),
                                _ptr(None if bias is None else bias.detach().contiguous()), _ptr(gate), _ptr(residual),
                                _ptr(y), B, Ca, cb, cb_src, cout, H, W, ks, _stream()), "wm_conv2d_fwd")
    return y


_F16_WS = {}


def _f16_ws_bytes(lib, cout, cin, ks):
    key = (cout, cin, ks)
    n = _F16_WS.get(key)
    if n is None:
        f = int(lib.wm_conv2d_wfrag_bytes(cout, cin, ks))
        n = _F16_WS[key] = f + 256 if f else 0
    return n


def conv2d_f16(x, weight, bias=None, dgrad=False):
    """y = F.conv2d(x, weight, bias, stride=1, padding=ks // 2), ks in {1, 3}, NCHW fp32, on the fp16 matrix cores with a
    two-term split of both operands and per-tensor power-of-two scales (csrc/conv2d.hip.h): ~1e-7 relative to the fp64
    result - the training step's form (forward, and the input gradient on the transposed, flipped weight).  Four launches:
    memset + the two largest magnitudes, weight fragments, convolution; nothing synchronises the host.  Forward only.
    dgrad: `weight` is the (Cin, Cout, ks, ks) weight of the FORWARD convolution whose input gradient this call computes from
    x = gy: the convolution with weight.transpose(0, 1).flip(2, 3), whose fragments the library reads from `weight`
    itself (autograd's formula: a flip and a copy kernel per convolution and step)."""
    lib = _lib.load()
    _require_cuda("conv2d_f16", x, weight, bias)
    B, Cin, H, W = x.shape
    cout, cin, ks = weight.shape[0], weight.shape[1], weight.shape[2]
    if dgrad:
        cout, cin = cin, cout
    if weight.dim() != 4 or weight.shape[3] != ks or ks not in (1, 3) or cin != Cin:
        raise RuntimeError(f"conv2d_f16: weight {tuple(weight.shape)} does not fit x {tuple(x.shape)} (ks in (1, 3))")
    if x.dtype != torch.float32 or weight.dtype != torch.float32:
        raise RuntimeError("conv2d_f16: float32 only")
    x = x.contiguous()
    w = weight.detach().contiguous()
    nws = _f16_ws_bytes(lib, cout, cin, ks)
    if nws == 0:
        check(_lib.WM_EUNSUPPORTED, "wm_conv2d_f16_steps")
    # wm_conv2d_f16_steps: the two maxima in a slot of the zeroed arena (no memset node), magnitudes + fragments in one launch
    amax = _zeros_small(2, x.device)
    wfrag = torch.empty(nws - 256, dtype=torch.uint8, device=x.device)
    y = torch.empty((B, cout, H, W), dtype=torch.float32, device=x.device)
    b = None if bias is None else bias.detach().contiguous()
    with torch.cuda.device(x.device):
        check(lib.wm_conv2d_f16_steps(_ptr(x), _ptr(w), _ptr(b), _ptr(y), _ptr(amax), _ptr(wfrag), B, cin, cout, H, W, ks,
                                      1 if dgrad else 0, _stream()), "wm_conv2d_f16_steps")
    return y


def conv2d_ln(x, ln_weight, ln_bias, ln_eps, weight, bias=None, residual=None):
    """conv1x1(LayerNorm2d(x)) + bias (+ residual) in one kernel (wm_conv2d_ln_fwd): x (B, 32, H, W) fp32, weight (Cout, 32, 1, 1).
    Bit-identical to layernorm2d() + conv2d().  Forward only."""
    lib = _lib.load()
    _require_cuda("conv2d_ln", x, ln_weight, ln_bias, weight, bias, residual)
    B, C, H, W = x.shape
    cout = weight.shape[0]
    if C != 32 or tuple(weight.shape[1:]) != (32, 1, 1) or x.dtype != torch.float32:
        raise NotImplementedError("conv2d_ln: fp32, 32 input channels, 1x1 weight")
    if residual is not None and tuple(residual.shape) != (B, cout, H, W):
        raise RuntimeError(f"conv2d_ln: residual must be {(B, cout, H, W)}, got {tuple(residual.shape)}")
    x = x.contiguous()
    frag = _conv2d_wfrag(weight)
    y = torch.empty((B, cout, H, W), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(lib.wm_conv2d_ln_fwd(_ptr(x), _ptr(_w(ln_weight)), _ptr(_w(ln_bias)), float(ln_eps), _ptr(frag),
                                   _ptr(None if bias is None else bias.detach().contiguous()),
                                   _ptr(None if residual is None else residual.contiguous()), _ptr(y), B, C, cout, H, W, _stream()),
              "wm_conv2d_ln_fwd")
    return y


_FUSE_LN_CONV = True      # False (tests / tools): LayerNorm2d and the 1x1 convolution as two launches


def patchify_conv_supported(img, weight, r):
    """Shapes wm_patchify_conv_fwd covers: nn.PixelUnshuffle(r) + 1x1 nn.Conv2d on an fp32 NCHW image."""
    return (img.is_cuda and img.dtype == torch.float32 and img.dim() == 4 and weight.dim() == 4 and r in (2, 4, 8)
            and tuple(weight.shape[2:]) == (1, 1) and weight.shape[1] == img.shape[1] * r * r and weight.shape[0] in (16, 32, 48, 64)
            and img.shape[2] % r == 0 and img.shape[3] % r == 0 and weight.shape[0] * weight.shape[1] <= 16384)


def patchify_conv(img, weight, bias, r):
    """conv1x1(pixel_unshuffle(img, r)) (the UNet's ps_down1..3, reference wavemamba_arch.py:1014-1025 / :1043-1045) as one
    r x r / stride-r convolution read straight from the image: the unshuffled tensor is never materialised.  Forward only.
    img (B, Cin, H, W) fp32; weight (Cout, Cin r r, 1, 1); bias (Cout) or None -> (B, Cout, H / r, W / r) fp32."""
    lib = _lib.load()
    _require_cuda("patchify_conv", img, weight)
    if not patchify_conv_supported(img, weight, r):
        raise ValueError("patchify_conv: unsupported shapes (see patchify_conv_supported)")
    img = img.contiguous()
    B, Cin, H, W = img.shape
    Cout = weight.shape[0]
    y = torch.empty((B, Cout, H // r, W // r), dtype=torch.float32, device=img.device)
    with torch.cuda.device(img.device):
        check(lib.wm_patchify_conv_fwd(_ptr(img), _ptr(_w(weight)), None if bias is None else _ptr(_w(bias)), _ptr(y), B, Cin, Cout,
                                       H, W, r, _stream()), "wm_patchify_conv_fwd")
    return y


def conv2d_ln_supported(x, weight):
    return (_FUSE_LN_CONV and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 32 and weight.dim() == 4
            and tuple(weight.shape[1:]) == (32, 1, 1) and x.shape[2] * x.shape[3] < 2 ** 31)


def conv2d_gated(x, weight3, weight1, bias1=None, x2=None, x2_index=None):
    """conv3x3(X; weight3) * sigmoid(conv1x1(X; weight1) + bias1) with X as in `conv2d` - PAConv's k3(x) * sigmoid(k2(x))
    (reference wavemamba_arch.py:694-697) in one kernel.  weight3 (Cout, Cin, 3, 3) without bias, weight1
    (Cout, Cin, 1, 1).  Forward only."""
    lib = _lib.load()
    _require_cuda("conv2d_gated", x, weight3, weight1, bias1, x2, x2_index)
    B, Ca, H, W = x.shape
    cout, cin = weight3.shape[0], weight3.shape[1]
    if tuple(weight3.shape) != (cout, cin, 3, 3) or tuple(weight1.shape) != (cout, cin, 1, 1):
        raise RuntimeError(f"conv2d_gated: weights must be (Cout, Cin, 3, 3) and (Cout, Cin, 1, 1), got "
                           f"{tuple(weight3.shape)} and {tuple(weight1.shape)}")
    cb = cb_src = 0
    if x2 is not None:
        if (x2.shape[0], x2.shape[2], x2.shape[3]) != (B, H, W):
            raise RuntimeError(f"conv2d_gated: x2 {tuple(x2.shape)} does not match x {tuple(x.shape)}")
        cb_src = x2.shape[1]
        cb = cb_src if x2_index is None else x2_index.shape[1]
    if cin != Ca + cb:
        raise RuntimeError(f"conv2d_gated: weights expect {cin} input channels, got {Ca} + {cb}")
    for t in (x, weight3, weight1, x2):
        if t is not None and t.dtype != torch.float32:
            raise RuntimeError("conv2d_gated: float32 only")
    x = x.contiguous()
    x2 = None if x2 is None else x2.contiguous()
    idx = None if x2_index is None else x2_index.to(torch.int32).contiguous()
    f3, f1 = _conv2d_wfrag(weight3), _conv2d_wfrag(weight1)
    y = torch.empty((B, cout, H, W), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(lib.wm_conv2d_gated_fwd(_ptr(x), _ptr(x2), _ptr(idx), _ptr(f3), _ptr(f1),
                                      _ptr(None if bias1 is None else bias1.detach().contiguous()), _ptr(y),
                                      B, Ca, cb, cb_src, cout, H, W, _stream()), "wm_conv2d_gated_fwd")
    return y


_LINEAR_WGRAD_SHAPES = {(128, 32), (32, 64), (64, 16), (16, 32), (32, 16), (16, 16), (64, 32), (32, 32), (16, 64)}


class _LinearNoBias(torch.autograd.Function):
    """F.linear(x, weight) for token tensors (..., I): ATen GEMMs for the output and the input gradient, the HIP MFMA
    reduction for the weight gradient (a (O, I) result with K = number of tokens)."""

    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        return F.linear(x, weight)

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x, weight = ctx.saved_tensors
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = gy.matmul(weight)
        if ctx.needs_input_grad[1]:
            O, I = weight.shape
            g2, x2 = gy.reshape(-1, O).contiguous().float(), x.reshape(-1, I).contiguous().float()
            gw = _zeros_small(O * I, x.device).view(O, I)
            with torch.cuda.device(x.device):
                check(lib.wm_linear_wgrad(_ptr(g2), _ptr(x2), _ptr(gw), g2.shape[0], O, I, _stream()), "wm_linear_wgrad")
        return gx, gw


def linear_nobias_supported(x, weight):
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32
            and tuple(weight.shape) in _LINEAR_WGRAD_SHAPES)


def linear_nobias(x, weight):
    """F.linear(x, weight) with the weight gradient on the HIP reduction kernel (training)."""
    _require_cuda("linear_nobias", x, weight)
    return _LinearNoBias.apply(x, weight)


def plane_sums(x):
    """x (B, C, H, W) fp32 -> (C,) sums over batch and plane (bias gradient of a convolution)."""
    lib = _lib.load()
    _require_cuda("plane_sums", x)
    B, C, H, W = x.shape
    x = x.contiguous().float()
    out = _zeros_small(C, x.device)
    with torch.cuda.device(x.device):
        check(lib.wm_plane_sums(_ptr(x), _ptr(out), B, C, H, W, _stream()), "wm_plane_sums")
    return out


class _L1Mean(torch.autograd.Function):
    """mean |a - b| of two dense fp32 tensors of one shape (nn.L1Loss() and the reduction of FFTLoss, femasr_model.py:171-179;
    losses.py:306-313): forward and backward on the HIP kernels of loss.hip.h.  (ATen's reduction zeroes its semaphores with
    hipMemsetAsync - a memset node in a captured training step, which this runtime must not be given: the reported loss of a
    replay went wrong, profiles/r06/graph_memset_node.md.)"""

    @staticmethod
    def forward(ctx, a, b):
        lib = _lib.load()
        ctx.save_for_backward(a, b)
        out = _zeros_small(1, a.device)
        with torch.cuda.device(a.device):
            check(lib.wm_l1_mean_fwd(_ptr(a), _ptr(b), _ptr(out), a.numel(), _stream()), "wm_l1_mean_fwd")
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        a, b = ctx.saved_tensors
        g = g.contiguous().float()
        ga = torch.empty_like(a)
        with torch.cuda.device(a.device):
            check(lib.wm_l1_mean_bwd(_ptr(a), _ptr(b), _ptr(g), _ptr(ga), a.numel(), _stream()), "wm_l1_mean_bwd")
        return (ga if ctx.needs_input_grad[0] else None), (-ga if ctx.needs_input_grad[1] else None)


def l1_mean(a, b):
    """F.l1_loss(a, b) (mean reduction) for CUDA fp32 tensors of one shape, differentiable in both arguments."""
    _require_cuda("l1_mean", a, b)
    if a.shape != b.shape:
        raise RuntimeError(f"l1_mean: shapes differ: {tuple(a.shape)} vs {tuple(b.shape)}")
    if a.numel() == 0:
        raise RuntimeError("l1_mean: empty input")
    return _L1Mean.apply(a.contiguous().float(), b.contiguous().float())


class _Conv2dTrain(torch.autograd.Function):
    """Dense 3x3 / 1x1 convolution (stride 1, 'same' padding) for training: forward and input gradient on the
    matrix-core kernels (the input gradient is the same convolution with the weight transposed and flipped) - the fp16 form
    (f16, conv2d_f16) or the inference kernels' split-bf16 form; weight and bias gradients from _conv_weight_grad / plane_sums."""

    @staticmethod
    def forward(ctx, x, weight, bias, f16):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.f16 = f16
        if f16:
            return conv2d_f16(x.detach(), weight.detach(), None if bias is None else bias.detach())
        return conv2d(x.detach(), weight.detach(), None if bias is None else bias.detach())

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous().float()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            if ctx.f16:
                gx = conv2d_f16(gy, weight, dgrad=True)                           # the transposed, flipped weight is never built
            else:
                wt = weight.detach().transpose(0, 1).flip(2, 3).contiguous()      # (Cin, Cout, ks, ks)
                gx = conv2d(gy, wt, None, dynamic_weight=True)
        gw, gb = _conv_param_grads(gy, x, weight, ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2])
        return gx, gw, gb, None


# Weight gradients of the dense convolutions in training: the HIP kernel (conv_wgrad.hip.h: K = positions on the bf16 matrix
# cores, split operands) instead of MIOpen's NHWC implicit-GEMM kernels and their layout transposes / hipBLASLt for 1x1.
# A leaf gradient - its ~4e-6 per product propagates nowhere.  set_train_conv_wgrad_hip(False) keeps ATen's.
_TRAIN_CONV_WGRAD_HIP = True


def set_train_conv_wgrad_hip(on):
    global _TRAIN_CONV_WGRAD_HIP
    prev, _TRAIN_CONV_WGRAD_HIP = _TRAIN_CONV_WGRAD_HIP, bool(on)
    return prev


def conv2d_wgrad_supported(x, weight):
    Cout, Cin, ks, ks2 = weight.shape
    return (x.is_cuda and x.dtype == torch.float32 and ks == ks2 and ks in (1, 3) and x.shape[3] % 32 == 0
            and (Cout + 15) // 16 in (1, 2, 4, 6) and not (ks == 3 and 64 < Cout <= 80))


def conv2d_wgrad(gy, x, ks, _refusal=RuntimeError, with_bias=False):
    """dW (Cout, Cin, ks, ks) of a stride-1 'same' convolution from gy (B, Cout, H, W) and x (B, Cin, H, W); with_bias: -> (dW, db),
    db (Cout) = gy summed over batch and plane, from the same pass over gy."""
    lib = _lib.load()
    _require_cuda("conv2d_wgrad", gy, x)
    if x.dim() != 4 or gy.dim() != 4 or gy.shape[0] != x.shape[0] or gy.shape[2:] != x.shape[2:]:
        raise _refusal(f"wm_conv2d_wgrad: gy {tuple(gy.shape)} does not match x {tuple(x.shape)}")
    B, Cout, H, W = gy.shape
    Cin = x.shape[1]
    gy = gy.contiguous().float(); x = x.contiguous().float()
    need = lib.wm_conv2d_wgrad_workspace_bytes(B, Cin, Cout, H, W, ks)
    if need == 0:
        raise _refusal("wm_conv2d_wgrad: unsupported shape (see conv2d_wgrad_supported)")
    ws = torch.empty(need, dtype=torch.uint8, device=x.device)
    dW = torch.empty(Cout, Cin, ks, ks, dtype=torch.float32, device=x.device)
    db = torch.empty(Cout, dtype=torch.float32, device=x.device) if with_bias else None
    with torch.cuda.device(x.device):
        rc = lib.wm_conv2d_wgrad(_ptr(gy), _ptr(x), _ptr(dW), _ptr(db), _ptr(ws), need, B, Cin, Cout, H, W, ks, _stream())
    if rc in (_lib.WM_EUNSUPPORTED, _lib.WM_EALIGN) and _refusal is not RuntimeError:
        raise _refusal(f"wm_conv2d_wgrad refused the call (code {rc})")
    check(rc, "wm_conv2d_wgrad")
    return (dW, db) if with_bias else dW


class _WgradRefused(RuntimeError):
    pass


def _conv_weight_grad(gy, x, weight, with_bias=False):
    """HIP matrix-core weight gradient, or ATen's for anything the kernel refuses - by the Python predicate, by
    wm_conv2d_wgrad_workspace_bytes (its shape limits) or by the call itself (WM_EUNSUPPORTED / WM_EALIGN): a training step
    never dies on a shape, as conv_wgrad.hip.h promises.  with_bias: -> (dW, db), the bias gradient from the same kernel (or
    plane_sums next to ATen's weight gradient)."""
    ks = weight.shape[2]
    if _TRAIN_CONV_WGRAD_HIP and conv2d_wgrad_supported(x, weight):
        try:
            return conv2d_wgrad(gy, x, ks, _refusal=_WgradRefused, with_bias=with_bias)
        except _WgradRefused:
            pass
    _, gw, _ = torch.ops.aten.convolution_backward(
        gy, x, weight, None, [1, 1], [ks // 2, ks // 2], [1, 1], False, [0, 0], 1, [False, True, False])
    return (gw, plane_sums(gy)) if with_bias else gw


def _conv_param_grads(gy, x, weight, want_w, want_b):
    """(dW, db) of a convolution's parameters, each None when not wanted: one kernel for both when both are."""
    if want_w and want_b:
        return _conv_weight_grad(gy, x, weight, with_bias=True)
    return (_conv_weight_grad(gy, x, weight) if want_w else None), (plane_sums(gy) if want_b else None)


class _Conv2dAten(torch.autograd.Function):
    """ATen's fp32 convolution (stride 1, 'same' padding) and input gradient, with the BIAS gradient on the HIP plane-sum
    kernel (autograd's own `grad_output.sum((0, 2, 3))` is ATen's generic strided reduction - 141 launches, 5.4 ms of a
    95-ms BASELINE config-3 training step, tools/train_breakdown.py) and the WEIGHT gradient on the HIP matrix-core
    kernel (`_conv_weight_grad`)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return F.conv2d(x, weight, bias, stride=1, padding=weight.shape[2] // 2)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        ks = weight.shape[2]
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx, _, _ = torch.ops.aten.convolution_backward(
                gy, x, weight, None, [1, 1], [ks // 2, ks // 2], [1, 1], False, [0, 0], 1, [True, False, False])
        gw, gb = _conv_param_grads(gy, x, weight, ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        return gx, gw, gb


# Training convolutions (forward + input gradient; weight / bias gradients: _conv_weight_grad, plane_sums).  Modes:
#   "auto"    (default) per call, whichever of the next two is faster at that shape (tools/bench_conv_train.py on MI355X,
#             profiles/r04/bench_conv_train.txt): the fp16 form for 3x3 convolutions over >= 2^17 positions and 1x1 over >= 2^19
#             (BASELINE config 3's level-1 / level-2 and full-resolution maps: 3x3 64 -> 64 at 8 x 256 x 256 0.137 against
#             0.371 ms, 32 -> 96 0.130 / 0.306, 64 -> 64 at 128 x 128 0.066 / 0.099, 1x1 32 -> 64 0.066 / 0.119), ATen below that
#             (8 x 64 x 64 maps: three launches cost more than MIOpen's one: 0.055 against 0.032 ms);
#   "f16"     always the fp16 form of the matrix-core kernels: two-term split, per-tensor power-of-two scales - 22 significant
#             bits per operand, fp32-class (conv2d_f16): a training step without MIOpen / hipBLASLt in it;
#   "aten"    always ATen's fp32 convolutions (MIOpen Winograd / hipBLASLt): the mode of rounds 3-4a, kept as the cross-check;
#   "bf16x3"  the inference kernels' split-bf16 form: 16 significant bits per operand (3-4e-6 on an output) - right for
#             inference (the contract's bar is 1e-4 on outputs), but in the training step those 3e-6 on the activations become
#             ~1e-4 on the parameter gradients of the deepest block, whose 8 x 8 maps make every gradient a short cancelling sum
#             (tools/grad_localize.py: prediction 2.8e-6 / worst gradient tensor 1.1e-4 with it, 7.7e-8 / 1.4e-5 with fp32
#             convolutions; the reference's own fp32: 9.7e-8 / 6e-6).  Never the default.
# WM_TRAIN_CONV=<mode> in the environment, or set_train_conv_mode().
_TRAIN_CONV_MODES = ("auto", "f16", "aten", "bf16x3")
_TRAIN_CONV_MODE = os.environ.get("WM_TRAIN_CONV", "auto")
if _TRAIN_CONV_MODE not in _TRAIN_CONV_MODES:
    raise RuntimeError(f"WM_TRAIN_CONV={_TRAIN_CONV_MODE!r}: one of {_TRAIN_CONV_MODES}")


def set_train_conv_mode(mode):
    """Select how autograd's dense convolutions run (see above).  Returns the previous mode."""
    global _TRAIN_CONV_MODE
    if mode not in _TRAIN_CONV_MODES:
        raise ValueError(f"set_train_conv_mode: {mode!r} is not one of {_TRAIN_CONV_MODES}")
    prev, _TRAIN_CONV_MODE = _TRAIN_CONV_MODE, mode
    return prev


def train_conv_mode():
    return _TRAIN_CONV_MODE


def conv2d_train(x, weight, bias=None):
    """F.conv2d(x, weight, bias, stride=1, padding=ks // 2) with autograd, in the mode set_train_conv_mode() selects."""
    _require_cuda("conv2d_train", x, weight, bias)
    mode = _TRAIN_CONV_MODE
    if mode == "auto":
        mode = "f16" if x.shape[0] * x.shape[2] * x.shape[3] >= ((1 << 17) if weight.shape[2] == 3 else (1 << 19)) else "aten"
        if mode == "f16" and _f16_ws_bytes(_lib.load(), weight.shape[0], weight.shape[1], weight.shape[2]) == 0:
            mode = "aten"                        # a shape the fp16 kernels refuse: the step never dies on it
    if mode == "aten":
        if not ((bias is not None and bias.requires_grad) or weight.requires_grad):
            return F.conv2d(x.float(), weight, bias, stride=1, padding=weight.shape[2] // 2)
        return _Conv2dAten.apply(x.float(), weight, bias)
    return _Conv2dTrain.apply(x.contiguous().float(), weight, bias, mode == "f16")


def conv2d_supported(x, weight, x2=None):
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 4
            and weight.dim() == 4 and tuple(weight.shape[2:]) in ((3, 3), (1, 1)) and x.shape[2] * x.shape[3] < 2 ** 31
            and (x2 is None or (x.shape[1] % 8 == 0 and x2.dtype == torch.float32)))


# ------------------------------------------------------------------------------------------------
# profiling hooks (bench.py)
# ------------------------------------------------------------------------------------------------
PROF_KERNELS = ("haar_analysis", "haar_synthesis", "selscan_chunk_reduce", "selscan_carry",
                "selscan_chunk_scan", "lfss_in", "ss2d_proj", "dwconv3x3",
                "ss2d_core_scan", "lfss_mid", "ss2d_core_reduce", "lfss_out", "selscan_bwd",
                "conv3x3", "conv1x1", "skff", "layernorm2d", "dwconv3x3_other", "patchify_conv", "unused_19")


def prof_enable(on=True):
    """on: False/True (every kernel class) or an iterable of PROF_KERNELS names (only those classes)."""
    if on is True:
        mask = 0xFFFFFFFF
    elif not on:
        mask = 0
    else:
        mask = 0
        for name in on:
            mask |= 1 << PROF_KERNELS.index(name)
    _lib.load().wm_prof_enable(mask)


def prof_collect():
    """-> {kernel name: (launches, total_ms)} since prof_enable(True); blocks on the recorded events."""
    import ctypes
    lib = _lib.load()
    n = (ctypes.c_int * _lib.WM_PROF_NKERNELS)()
    ms = (ctypes.c_double * _lib.WM_PROF_NKERNELS)()
    check(lib.wm_prof_collect(n, ms), "wm_prof_collect")
    return {PROF_KERNELS[k]: (n[k], ms[k]) for k in range(_lib.WM_PROF_NKERNELS)}
