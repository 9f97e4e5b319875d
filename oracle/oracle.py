"""Python face of the CPU oracle (oracle/wavemamba_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this module; the
product package (wave_mamba_amd/) never does.  It exposes the same operator names as
wave_mamba_amd.ops (dwt_init, iwt_init, iwt_init_pair, selective_scan_fn) on CPU tensors, backed by
the plain-C restatement of the reference algorithms, so the very same network code can be run and
timed on host cores as the checker / CPU baseline.
"""
import ctypes
import os
import subprocess

import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "wavemamba_oracle.c")
LIB = os.path.join(HERE, "libwavemamba_oracle.so")

_lib = None


def build(force=False):
    """gcc -O3 -fopenmp (baseline x86-64, no -march: the .so travels to the GPU box's host CPU)."""
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        cmd = ["gcc", "-O3", "-fopenmp", "-fPIC", "-shared", "-ffp-contract=off", "-std=c11", SRC,
               "-o", LIB + ".tmp", "-lm"]
        subprocess.run(cmd, check=True)
        os.replace(LIB + ".tmp", LIB)
    return LIB


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(LIB)
        _lib.oracle_num_threads.restype = ctypes.c_int
    return _lib


def num_threads():
    return lib().oracle_num_threads()


def usable_cpus(cap=64):
    """Host cores this process may really use: affinity mask and cgroup CPU quota, capped (a 256-way
    OpenMP team on a quota-limited container spends its time spinning, not computing)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return max(1, min(n, cap))


def set_num_threads(n):
    lib().oracle_set_num_threads(int(n))


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _f32(t):
    return None if t is None else t.detach().contiguous().float()


# ---- raw (no autograd) entry points ---------------------------------------------------------------
def dwt_raw(x):
    x = _f32(x)
    B, C, H, W = x.shape
    if H % 2 or W % 2:
        raise RuntimeError(f"dwt_init: H and W must be even, got {H}x{W}")
    outs = [torch.empty(B, C, H // 2, W // 2) for _ in range(4)]
    lib().oracle_dwt2d_fwd(_p(x), *[_p(o) for o in outs], B, C, H, W)
    return tuple(outs)


def iwt_raw(x):
    x = _f32(x)
    B, C4, h, w = x.shape
    out = torch.empty(B, C4 // 4, 2 * h, 2 * w)
    lib().oracle_idwt2d_fwd(_p(x), _p(out), B, C4 // 4, h, w)
    return out


def selscan_fwd_raw(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                    return_last_state=False):
    u, delta, A, B, C, D, z, delta_bias = [_f32(t) for t in (u, delta, A, B, C, D, z, delta_bias)]
    if B.dim() == 3:
        B = B.unsqueeze(1).contiguous()
    if C.dim() == 3:
        C = C.unsqueeze(1).contiguous()
    batch, dim, L = u.shape
    N, G = A.shape[1], B.shape[1]
    out = torch.empty_like(u)
    last = torch.empty(batch, dim, N) if return_last_state else None
    lib().oracle_selscan_fwd(_p(u), _p(delta), _p(A), _p(B), _p(C), _p(D), _p(z), _p(delta_bias),
                             _p(out), _p(last), batch, dim, L, N, G, int(bool(delta_softplus)))
    return (out, last) if return_last_state else out


def selscan_bwd_raw(u, delta, A, B, C, D, delta_bias, dy, delta_softplus=True):
    u, delta, A, B, C, D, delta_bias, dy = [_f32(t) for t in (u, delta, A, B, C, D, delta_bias, dy)]
    batch, dim, L = u.shape
    N, G = A.shape[1], B.shape[1]
    du, dd = torch.empty_like(u), torch.empty_like(delta)
    dA, dB, dC = torch.zeros_like(A), torch.zeros_like(B), torch.zeros_like(C)
    dD = torch.zeros(dim)
    dbias = torch.zeros(dim)
    lib().oracle_selscan_bwd(_p(u), _p(delta), _p(A), _p(B), _p(C), _p(D), _p(delta_bias), _p(dy),
                             _p(du), _p(dd), _p(dA), _p(dB), _p(dC), _p(dD), _p(dbias),
                             batch, dim, L, N, G, int(bool(delta_softplus)))
    return du, dd, dA, dB, dC, (dD if D is not None else None), (dbias if delta_bias is not None else None)


def ss2d_core_raw(x, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds):
    x, Wx, Wdt, bias, Al, Dv = [_f32(t) for t in (x, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds)]
    B, D, H, W = x.shape
    N = Al.shape[1]
    R = Wdt.shape[2]
    y = torch.empty(4, B, D, H * W)
    lib().oracle_ss2d_core_fwd(_p(x), _p(Wx), _p(Wdt), _p(bias), _p(Al), _p(Dv), _p(y), B, D, H, W, N, R)
    return y[0], y[1], y[2], y[3]


# ---- autograd-capable operator surface (same names as wave_mamba_amd.ops) ---------------------------
class _DWT(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.dtype = x.dtype
        return tuple(o.to(x.dtype) for o in dwt_raw(x))

    @staticmethod
    def backward(ctx, *gs):
        # the scaled Haar matrix is orthogonal: d(analysis) = synthesis
        return iwt_raw(torch.cat([g.float() for g in gs], dim=1)).to(ctx.dtype)


class _IWT(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.dtype = x.dtype
        return iwt_raw(x)

    @staticmethod
    def backward(ctx, g):
        return torch.cat(dwt_raw(g), dim=1).to(ctx.dtype)


def dwt_init(x):
    return _DWT.apply(x)


def iwt_init(x):
    return _IWT.apply(x)


def iwt_init_pair(x_l, x_h):
    return _IWT.apply(torch.cat([x_l, x_h], dim=1))


class _Scan(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, delta, A, B, C, D, delta_bias, delta_softplus):
        ctx.save_for_backward(u, delta, A, B, C, D, delta_bias)
        ctx.sp = delta_softplus
        return selscan_fwd_raw(u, delta, A, B, C, D, None, delta_bias, delta_softplus)

    @staticmethod
    def backward(ctx, dy):
        u, delta, A, B, C, D, bias = ctx.saved_tensors
        return (*selscan_bwd_raw(u, delta, A, B, C, D, bias, dy, ctx.sp), None)


def selective_scan_fn(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                      return_last_state=False):
    in_dtype = u.dtype
    B4 = B.unsqueeze(1) if B.dim() == 3 else B
    C4 = C.unsqueeze(1) if C.dim() == 3 else C
    need_grad = torch.is_grad_enabled() and any(
        t is not None and t.requires_grad for t in (u, delta, A, B, C, D, z, delta_bias))
    if need_grad:
        assert not return_last_state
        out = _Scan.apply(u.float(), delta.float(), A.float(), B4.float().contiguous(),
                          C4.float().contiguous(), D, delta_bias, bool(delta_softplus))
        if z is not None:
            out = out * F.silu(z.float())
        return out.to(in_dtype)
    res = selscan_fwd_raw(u, delta, A, B4, C4, D, z, delta_bias, delta_softplus, return_last_state)
    if return_last_state:
        return res[0].to(in_dtype), res[1]
    return res.to(in_dtype)
