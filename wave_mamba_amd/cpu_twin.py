"""CPU twins of the hot-path operators of the drop-in boundary ("each with CPU + HIP implementations"; BASELINE config 1: the
LOLv1 / 1x3x256x256 "CPU-only PyTorch forward (plumbing, no GPU)").

    dwt_init            Haar analysis               /root/reference/basicsr/archs/wavemamba_arch.py:97-110
    iwt_init(_pair)     Haar synthesis              :113-130
    selective_scan_fn   the selective-scan operator :465-471 (mamba_ssm's published selective_scan_ref semantics)
    ss2d_core           SS2D.forward_core           :446-478 (for the torch.library op wavemamba_hip::ss2d_core on CPU tensors)

Plain PyTorch, differentiable by autograd, written from the operators' definitions.  They are
the implementation for CPU TENSORS and nothing else: `ops.py` selects them by the device of the input, never by the absence of
the HIP library - a CUDA (ROCm) tensor always runs the HIP kernels and raises when the library is missing
(tests/test_cabi.py::test_missing_library_fails_loudly, tests/test_cpu_twin.py::test_other_devices_never_reach_the_cpu_twins,
tests/test_gpu_parity.py::test_cuda_tensors_fail_loudly_without_the_library).
Nothing here imports `oracle/` (the C restatement the tests check BOTH implementations against), and bench.py /
__graft_entry__.smoke() never run it.  Speed is not a goal: the scan walks the sequence step by step.
"""
import torch
import torch.nn.functional as F


def _quads(x):
    """The four polyphase components of an even-sized NCHW map, each pre-divided by 2 (:98-103):
    a = x[2i, 2j], b = x[2i+1, 2j], c = x[2i, 2j+1], d = x[2i+1, 2j+1]."""
    if x.dim() != 4 or x.shape[2] % 2 or x.shape[3] % 2:
        raise RuntimeError(f"dwt_init: expected (B, C, H, W) with even H and W, got {tuple(x.shape)}")
    even_rows, odd_rows = x[:, :, 0::2, :] / 2, x[:, :, 1::2, :] / 2
    return even_rows[..., 0::2], odd_rows[..., 0::2], even_rows[..., 1::2], odd_rows[..., 1::2]


def dwt_init(x):
    """(LL, HL, LH, HH), each (B, C, H/2, W/2), dtype of x (:104-110)."""
    a, b, c, d = _quads(x)
    return a + b + c + d, -a - b + c + d, -a + b - c + d, a - b - c + d


def iwt_init(x):
    """(B, 4C, h, w) = [x1 | x2 | x3 | x4] channel blocks -> (B, C, 2h, 2w), always float32 (:113-130)."""
    if x.dim() != 4 or x.shape[1] % 4:
        raise RuntimeError(f"iwt_init: expected (B, 4C, h, w), got {tuple(x.shape)}")
    B, C4, h, w = x.shape
    C = C4 // 4
    x1, x2, x3, x4 = (x[:, i * C:(i + 1) * C].float() / 2 for i in range(4))
    out = torch.empty(B, C, 2 * h, 2 * w, dtype=torch.float32, device=x.device)
    out[:, :, 0::2, 0::2] = x1 - x2 - x3 + x4
    out[:, :, 1::2, 0::2] = x1 - x2 + x3 - x4
    out[:, :, 0::2, 1::2] = x1 + x2 - x3 - x4
    out[:, :, 1::2, 1::2] = x1 + x2 + x3 + x4
    return out


def iwt_init_pair(x_l, x_h):
    return iwt_init(torch.cat([x_l, x_h], dim=1))


def selective_scan_fn(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False, return_last_state=False):
    """h_t = exp(dt_t A) h_{t-1} + dt_t B_t u_t,  y_t = C_t . h_t + D u_t  (then y * silu(z)), dt = softplus(delta + bias) when
    delta_softplus.  u, delta (batch, dim, L); A (dim, N); B, C (batch, [G,] N, L) - the G groups split `dim` evenly; D, delta_bias
    (dim).  fp32 arithmetic, `out` in u's dtype; the fp32 last state (batch, dim, N) when return_last_state."""
    if u.dim() != 3 or delta.shape != u.shape or A.dim() != 2 or A.shape[0] != u.shape[1]:
        raise RuntimeError(f"selective_scan_fn: inconsistent shapes u {tuple(u.shape)}, delta {tuple(delta.shape)}, A {tuple(A.shape)}")
    batch, dim, L = u.shape
    N = A.shape[1]
    in_dtype = u.dtype
    u, dt, A = u.float(), delta.float(), A.float()
    if delta_bias is not None:
        dt = dt + delta_bias.float()[None, :, None]
    if delta_softplus:
        dt = F.softplus(dt)
    B4 = B.float() if B.dim() == 4 else B.float()[:, None]
    C4 = C.float() if C.dim() == 4 else C.float()[:, None]
    G = B4.shape[1]
    if B4.shape != (batch, G, N, L) or C4.shape != B4.shape or dim % G:
        raise RuntimeError(f"selective_scan_fn: B {tuple(B.shape)} / C {tuple(C.shape)} do not fit u {tuple(u.shape)}, A {tuple(A.shape)}")
    Bd = B4.repeat_interleave(dim // G, dim=1)               # (batch, dim, N, L): the group's B for each of its channels
    Cd = C4.repeat_interleave(dim // G, dim=1)
    h = u.new_zeros(batch, dim, N)
    ys = []
    for t in range(L):
        dt_t = dt[:, :, t, None]                             # (batch, dim, 1)
        h = torch.exp(dt_t * A) * h + (dt_t * u[:, :, t, None]) * Bd[:, :, :, t]
        ys.append((h * Cd[:, :, :, t]).sum(-1))
    y = torch.stack(ys, dim=-1) if ys else u.new_zeros(batch, dim, 0)
    if D is not None:
        y = y + u * D.float()[None, :, None]
    if z is not None:
        y = y * F.silu(z.float())
    y = y.to(in_dtype)
    return (y, h) if return_last_state else y


def ss2d_core(x, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds):
    """SS2D.forward_core (:446-478) on an NCHW map x (B, D, H, W): the four scan orders - row-major, column-major and their
    reversals (:451-452) - the per-direction projections x_dbl = x_proj_weight[k] . xs[k] -> (dt_r | B | C) and
    dts = dt_projs_weight[k] . dt_r (:453-455), one grouped selective scan (:465-471), and the four outputs brought back to
    row-major positions (:474-478).  Returns (y_row, y_row_reversed, y_col, y_col_reversed), each (B, D, H W) float32."""
    B, D, H, W = x.shape
    L = H * W
    K, N, R = 4, A_logs.shape[1], dt_projs_weight.shape[2]
    x = x.float()
    two = torch.stack([x.reshape(B, D, L), x.transpose(2, 3).reshape(B, D, L)], dim=1)
    xs = torch.cat([two, two.flip(-1)], dim=1)                                           # (B, 4, D, L)
    x_dbl = torch.einsum("bkdl,kcd->bkcl", xs, x_proj_weight.float())
    dt_r, Bs, Cs = torch.split(x_dbl, [R, N, N], dim=2)
    dts = torch.einsum("bkrl,kdr->bkdl", dt_r, dt_projs_weight.float())
    out = selective_scan_fn(xs.reshape(B, K * D, L), dts.reshape(B, K * D, L), -torch.exp(A_logs.float()), Bs, Cs, Ds.float(), None,
                            dt_projs_bias.float().reshape(-1), True).reshape(B, K, D, L)
    back = out[:, 2:4].flip(-1)
    to_rows = lambda t: t.reshape(B, D, W, H).transpose(2, 3).reshape(B, D, L)
    return out[:, 0], back[:, 0], to_rows(out[:, 1]), to_rows(back[:, 1])
