"""WaveMamba for MI355X: the reference's `ARCH_REGISTRY` entry, re-built on the HIP hot path.

Boundary contract (citations into /root/reference/basicsr/archs/wavemamba_arch.py):
  * registered under the name `WaveMamba` (:1066); ctor `WaveMamba(*, in_chn, wf, n_l_blocks,
    n_h_blocks, ffn_scale, **ignore_kwargs)` (:1068-1075); attribute `restoration_network`
    (:1077, used directly by inference_wavemamba.py:109); methods forward / test / test_tile /
    check_image_size / encode_and_decode / print_network (:1079-1176);
  * identical state-dict keys and shapes (591 for the shipped config), so reference-format
    checkpoints `torch.load(p)['params']` load unchanged (inference_wavemamba.py:77);
  * modules are created in the reference's order, so `torch.manual_seed(s)` + construction yields
    the reference's initial weights bit for bit (checked by tests against committed fingerprints).

Hot path (the part that is NOT PyTorch): the three Haar DWTs and three IWTs per forward
(DownFRG / upFRG) and the 14 LFSSBlocks (SS2D four-direction core, its prologue / epilogue, the gated ffn) go to
hand-written gfx950 kernels through `wave_mamba_amd.ops`; so does, as the first "next" row of SURVEY.md section 8f,
every full-map operator of the HFE branch and the U-Net plumbing in inference (dense 3x3 / 1x1 convolutions with their
concatenations, gathers, gates and residuals fused, Gram matrices, channel matching, the folded attention, SKFF,
LayerNorm2d, depth-wise convolutions).  Training takes the HIP kernels that have a backward (DWT / IWT, the SS2D core,
depth-wise conv, the LayerNorms) and PyTorch autograd for the rest.

This file can be dropped into a `basicsr/archs/` folder: the auto-scan (`archs/__init__.py:12-16`)
imports every `*_arch.py`; inside a `basicsr` tree the import below takes the real `basicsr.utils.registry.ARCH_REGISTRY`,
inside this package the package's own `registry.py` (two registries, never one name registered twice).
"""
import math

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

try:                                                   # inside the wave_mamba_amd package
    from .. import ops as _hip_ops
    from ..registry import ARCH_REGISTRY
except ImportError:                                    # dropped into basicsr/archs/
    import wave_mamba_amd.ops as _hip_ops
    from basicsr.utils.registry import ARCH_REGISTRY


class _OpsBackend:
    """The hot-path operators: the HIP library, always.  (The class attribute is the one place the test suite's CPU
    oracle patches from outside - oracle/backend.py - to run the same network on host cores; nothing in this package
    installs anything here.)"""
    impl = _hip_ops


def _needs_grad(module, *tensors):
    """Autograd will want gradients through this module: grad mode is on and an input or ANY parameter of the module
    requires grad (a block with a frozen skip_scale or frozen early layers still has trainable weights inside)."""
    if not torch.is_grad_enabled():
        return False
    if any(t is not None and t.requires_grad for t in tensors):
        return True
    return module is not None and any(q.requires_grad for q in module.parameters())


def _dwconv(conv, x, act="none"):
    """Depth-wise 3x3 nn.Conv2d (+ optional SiLU / exact GELU).  Inference on the HIP backend goes to the
    streaming HIP kernel; training (autograd) and the test backends use the PyTorch conv."""
    ops = _OpsBackend.impl
    if hasattr(ops, "dwconv3x3") and x.is_cuda and x.dtype == torch.float32:
        if not _needs_grad(conv, x):
            return ops.dwconv3x3(x, conv.weight, conv.bias, act)
        y = ops.dwconv3x3_train(x, conv.weight, conv.bias)          # HIP forward + backward (autograd)
    else:
        y = conv(x)
    return F.silu(y) if act == "silu" else F.gelu(y) if act == "gelu" else y


def _ln_tok(norm, x):
    """nn.LayerNorm over the last axis of token tensors: the HIP kernel pair (forward + backward) on the HIP backend,
    the module elsewhere."""
    ops = _OpsBackend.impl
    if (hasattr(ops, "layernorm_tok") and isinstance(norm, nn.LayerNorm) and norm.elementwise_affine
            and norm.bias is not None and len(norm.normalized_shape) == 1
            and ops.layernorm_tok_supported(x, norm.normalized_shape[0])):
        return ops.layernorm_tok(x, norm.weight, norm.bias, norm.eps)
    return norm(x)


def _linear(lin, x):
    """Bias-free nn.Linear on token tensors: in training on the HIP backend the weight gradient (a tiny matrix reduced
    over ~5e5 tokens) runs on the MFMA reduction kernel instead of a 32 x 32-tile library GEMM."""
    ops = _OpsBackend.impl
    if (lin.bias is None and hasattr(ops, "linear_nobias") and torch.is_grad_enabled() and lin.weight.requires_grad
            and ops.linear_nobias_supported(x, lin.weight)):
        return ops.linear_nobias(x, lin.weight)
    return lin(x)


def _conv(conv, x, x2=None, x2_index=None, gate=None, residual=None):
    """Dense 3x3 / 1x1 nn.Conv2d (stride 1, 'same' padding) of X = x | cat([x, x2], 1) |
    cat([x, gather(x2, 1, x2_index)], 1), then `* sigmoid(gate)` and `+ residual` when given.  Inference on the
    HIP backend is one matrix-core kernel (no concatenation, gather, bias, gate or residual kernels); training
    (autograd) and the test backends compose the PyTorch ops."""
    ops = _OpsBackend.impl
    if (hasattr(ops, "conv2d") and ops.conv2d_supported(x, conv.weight, x2)
            and not _needs_grad(conv, x, x2, gate, residual)):
        return ops.conv2d(x, conv.weight, conv.bias, x2, x2_index, gate, residual)
    x = _cat_gathered(x, x2, x2_index)
    if hasattr(ops, "conv2d_train") and ops.conv2d_supported(x, conv.weight) and torch.is_grad_enabled():
        y = ops.conv2d_train(x, conv.weight, conv.bias)             # ops.set_train_conv_mode(): fp16-split HIP kernels by default
    else:
        y = conv(x)
    if gate is not None:
        if (hasattr(ops, "gate_act") and torch.is_grad_enabled() and y.is_cuda and y.dtype == torch.float32
                and ops.gate_supported(gate, y)):
            y = ops.gate_act(gate, y, "sigmoid")                    # one launch forward, one backward (training)
        else:
            y = y * torch.sigmoid(gate)
    if residual is not None:
        y = y + residual
    return y


def _ps_conv(ps, img):
    """ps = nn.Sequential(nn.PixelUnshuffle(r), 1x1 nn.Conv2d) on the input image (reference :1014-1025, :1043-1045).  Inference
    on the HIP backend: one r x r / stride-r kernel reading the image itself (no unshuffled copy); otherwise the two modules."""
    ops = _OpsBackend.impl
    r, conv = ps[0].downscale_factor, ps[1]
    if (hasattr(ops, "patchify_conv") and ops.patchify_conv_supported(img, conv.weight, r) and not _needs_grad(conv, img)):
        return ops.patchify_conv(img, conv.weight, conv.bias, r)
    return _conv(conv, ps[0](img))


def _ln_conv(norm, conv, x):
    """conv(norm(x)) for a LayerNorm2d `norm` (or None) and a 1x1 nn.Conv2d: HFEBlock's norm1 -> attn.qkv and norm2 ->
    ffn.project_in[0] (reference :843-851).  Inference on the HIP backend: ONE kernel (the normalisation happens in the 1x1
    kernel's staging registers - bit-identical to the two launches, 256 B per position less); otherwise the two modules."""
    if norm is None:
        return _conv(conv, x)
    ops = _OpsBackend.impl
    if (hasattr(ops, "conv2d_ln") and ops.conv2d_ln_supported(x, conv.weight) and not _needs_grad(conv, x)
            and not _needs_grad(norm, x)):
        return ops.conv2d_ln(x, norm.weight, norm.bias, norm.eps, conv.weight, conv.bias)
    return _conv(conv, norm(x))


def _cat_gathered(x, x2=None, x2_index=None):
    """cat([x, gather(x2, 1, x2_index)], 1) (x2_index None: cat([x, x2], 1); x2 None: x)."""
    if x2 is None:
        return x
    if x2_index is not None:
        x2 = torch.gather(x2, 1, x2_index.long()[:, :, None, None].expand(-1, -1, x2.shape[2], x2.shape[3]))
    return torch.cat([x, x2], dim=1)


# ================================================================================================
# Low-frequency branch: SS2D / LFSSBlock
# ================================================================================================
class DWT(nn.Module):
    """Haar analysis (reference :133-139)."""

    def forward(self, x):
        return _OpsBackend.impl.dwt_init(x)


class IWT(nn.Module):
    """Haar synthesis (reference :142-148); also accepts the un-concatenated (x_l, x_h) pair."""

    def forward(self, x, x_h=None):
        if x_h is None:
            return _OpsBackend.impl.iwt_init(x)
        return _OpsBackend.impl.iwt_init_pair(x, x_h)


class ffn(nn.Module):
    """Gated depth-wise conv feed-forward of LFSSBlock (reference :214-231)."""

    def __init__(self, num_feat, ffn_expand=2):
        super().__init__()
        hidden = num_feat * ffn_expand
        self.conv1 = nn.Conv2d(num_feat, hidden, kernel_size=1)
        self.conv2 = nn.Conv2d(hidden, hidden, kernel_size=3, padding=1, groups=hidden)
        self.conv3 = nn.Conv2d(hidden // 2, num_feat, kernel_size=1)

    def forward(self, x):
        t = _dwconv(self.conv2, _conv(self.conv1, x))
        ops = _OpsBackend.impl
        if torch.is_grad_enabled() and t.requires_grad and t.is_cuda and t.dtype == torch.float32 and hasattr(ops, "glu_gate"):
            return _conv(self.conv3, ops.glu_gate(t, "gelu"))          # training: chunk + gate, one launch each way
        gate, value = t.chunk(2, dim=1)
        return _conv(self.conv3, F.gelu(gate) * value)


class SS2D(nn.Module):
    """Four-direction selective-scan block (reference :316-497).

    Parameter names, shapes and initialisers follow :345-386 / :389-444 exactly:
      in_proj.weight (2D_in, C) | conv2d.{weight (D_in,1,3,3), bias} | x_proj_weight (4, R+2N, D_in)
      dt_projs_weight (4, D_in, R) | dt_projs_bias (4, D_in) | A_logs (4 D_in, N) | Ds (4 D_in)
      out_norm.{weight,bias} (D_in) | out_proj.weight (C, D_in)
    """

    def __init__(self, d_model, d_state=16, d_conv=3, expand=2, dt_rank="auto", dt_min=0.001,
                 dt_max=0.1, dt_init="random", dt_scale=1.0, dt_init_floor=1e-4, dropout=0.0,
                 conv_bias=True, bias=False, device=None, dtype=None, **kwargs):
        super().__init__()
        fk = {"device": device, "dtype": dtype}
        self.d_model, self.d_state, self.d_conv, self.expand = d_model, d_state, d_conv, expand
        self.d_inner = int(expand * d_model)
        self.dt_rank = math.ceil(d_model / 16) if dt_rank == "auto" else dt_rank
        K = 4

        self.in_proj = nn.Linear(d_model, 2 * self.d_inner, bias=bias, **fk)
        self.conv2d = nn.Conv2d(self.d_inner, self.d_inner, kernel_size=d_conv, padding=(d_conv - 1) // 2,
                                groups=self.d_inner, bias=conv_bias, **fk)
        self.act = nn.SiLU()

        # per-direction projections, stacked into single parameters (:357-378)
        xp = [nn.Linear(self.d_inner, self.dt_rank + 2 * d_state, bias=False, **fk) for _ in range(K)]
        self.x_proj_weight = nn.Parameter(torch.stack([m.weight for m in xp], dim=0))
        dtp = [self.dt_init(self.dt_rank, self.d_inner, dt_scale, dt_init, dt_min, dt_max, dt_init_floor, **fk)
               for _ in range(K)]
        self.dt_projs_weight = nn.Parameter(torch.stack([m.weight for m in dtp], dim=0))
        self.dt_projs_bias = nn.Parameter(torch.stack([m.bias for m in dtp], dim=0))

        self.A_logs = self.A_log_init(d_state, self.d_inner, copies=K, merge=True)
        self.Ds = self.D_init(self.d_inner, copies=K, merge=True)

        self.out_norm = nn.LayerNorm(self.d_inner)
        self.out_proj = nn.Linear(self.d_inner, d_model, bias=bias, **fk)
        self.dropout = nn.Dropout(dropout) if dropout > 0.0 else None

    # ---- initialisers (:389-444) ------------------------------------------------------------
    @staticmethod
    def dt_init(dt_rank, d_inner, dt_scale=1.0, dt_init="random", dt_min=0.001, dt_max=0.1,
                dt_init_floor=1e-4, **fk):
        proj = nn.Linear(dt_rank, d_inner, bias=True, **fk)
        std = dt_rank ** -0.5 * dt_scale
        if dt_init == "constant":
            nn.init.constant_(proj.weight, std)
        elif dt_init == "random":
            nn.init.uniform_(proj.weight, -std, std)
        else:
            raise NotImplementedError
        # bias = softplus^-1(dt), dt log-uniform in [dt_min, dt_max]
        dt = torch.exp(torch.rand(d_inner, **fk) * (math.log(dt_max) - math.log(dt_min))
                       + math.log(dt_min)).clamp(min=dt_init_floor)
        with torch.no_grad():
            proj.bias.copy_(dt + torch.log(-torch.expm1(-dt)))
        proj.bias._no_reinit = True
        return proj

    @staticmethod
    def A_log_init(d_state, d_inner, copies=1, device=None, merge=True):
        A_log = torch.log(torch.arange(1, d_state + 1, dtype=torch.float32, device=device))
        A_log = A_log.repeat(d_inner, 1).contiguous()                 # S4D-real: A[d, n] = n + 1
        if copies > 1:
            A_log = A_log.unsqueeze(0).repeat(copies, 1, 1)
            if merge:
                A_log = A_log.flatten(0, 1)
        A_log = nn.Parameter(A_log)
        A_log._no_weight_decay = True
        return A_log

    @staticmethod
    def D_init(d_inner, copies=1, device=None, merge=True):
        D = torch.ones(d_inner, device=device)
        if copies > 1:
            D = D.unsqueeze(0).repeat(copies, 1)
            if merge:
                D = D.flatten(0, 1)
        D = nn.Parameter(D)
        D._no_weight_decay = True
        return D

    # ---- forward ----------------------------------------------------------------------------
    def forward_core(self, x):
        """x (B, D_in, H, W) -> four (B, D_in, L) tensors in row-major l (reference :446-478)."""
        ops = _OpsBackend.impl
        if self._fused_ok(x):
            return ops.ss2d_core(x, self.x_proj_weight, self.dt_projs_weight, self.dt_projs_bias,
                                 self.A_logs, self.Ds)
        B, D, H, W = x.shape
        L, K, N, R = H * W, 4, self.d_state, self.dt_rank
        row = x.reshape(B, D, L)
        col = x.transpose(2, 3).reshape(B, D, L)                      # l = w*H + h
        fwd = torch.stack([row, col], dim=1)                          # (B, 2, D, L)
        xs = torch.cat([fwd, fwd.flip(-1)], dim=1)                    # (B, 4, D, L)
        x_dbl = torch.einsum("bkdl,kcd->bkcl", xs, self.x_proj_weight)
        dts, Bs, Cs = torch.split(x_dbl, [R, N, N], dim=2)
        dts = torch.einsum("bkrl,kdr->bkdl", dts, self.dt_projs_weight)
        out = ops.selective_scan_fn(
            xs.float().reshape(B, K * D, L), dts.contiguous().float().reshape(B, K * D, L),
            -torch.exp(self.A_logs.float()), Bs.float(), Cs.float(), self.Ds.float(), z=None,
            delta_bias=self.dt_projs_bias.float().reshape(-1), delta_softplus=True,
            return_last_state=False).reshape(B, K, D, L)
        assert out.dtype == torch.float32
        back = out[:, 2:4].flip(-1)
        y_col = out[:, 1].reshape(B, D, W, H).transpose(2, 3).reshape(B, D, L)
        y_col_back = back[:, 1].reshape(B, D, W, H).transpose(2, 3).reshape(B, D, L)
        return out[:, 0], back[:, 0], y_col, y_col_back

    def _fused_ok(self, x):
        """The fused HIP core (forward wm_ss2d_core_fwd, backward wm_ss2d_core_bwd) serves inference and training
        on the HIP backend; out-of-range shapes and the test backends take the direction glue +
        selective_scan_fn path below."""
        ops = _OpsBackend.impl
        if not (hasattr(ops, "ss2d_core") and x.is_cuda and x.dtype == torch.float32):
            return False
        if not ops.ss2d_core_supported(self.d_inner, self.d_state, self.dt_rank, x.shape[-1], x.shape[-2]):
            return False
        return not _needs_grad(self, x) or ops.ss2d_core_bwd_supported(self.d_inner, self.d_state, self.dt_rank)

    def forward(self, x, **kwargs):
        B, H, W, C = x.shape
        x, z = _linear(self.in_proj, x).chunk(2, dim=-1)
        x = _dwconv(self.conv2d, x.permute(0, 3, 1, 2).contiguous(), act="silu")
        if self._fused_ok(x):                      # y1 + y2 + y3 + y4 accumulated inside the kernels
            y = _OpsBackend.impl.ss2d_core(x, self.x_proj_weight, self.dt_projs_weight, self.dt_projs_bias,
                                           self.A_logs, self.Ds, merged=True)
        else:
            y1, y2, y3, y4 = self.forward_core(x)
            assert y1.dtype == torch.float32
            y = y1 + y2 + y3 + y4
        y = y.transpose(1, 2).contiguous().view(B, H, W, -1)
        y = _ln_tok(self.out_norm, y) * F.silu(z)
        y = _linear(self.out_proj, y)
        return self.dropout(y) if self.dropout is not None else y


class LFSSBlock(nn.Module):
    """LN -> SS2D -> scaled skip; LN -> gated conv ffn -> scaled skip, on (B, HW, C) (:499-528)."""

    def __init__(self, hidden_dim=0, drop_path=0.0, norm_layer=None, attn_drop_rate=0.0, d_state=16,
                 expand=2.0, **kwargs):
        super().__init__()
        if drop_path:
            raise NotImplementedError("stochastic depth is unused by the reference (rate 0, :513)")
        self.ln_1 = norm_layer(hidden_dim) if norm_layer is not None else nn.LayerNorm(hidden_dim, eps=1e-6)
        self.self_attention = SS2D(d_model=hidden_dim, d_state=d_state, expand=expand,
                                   dropout=attn_drop_rate, **kwargs)
        self.drop_path = nn.Identity()
        self.skip_scale = nn.Parameter(torch.ones(hidden_dim))
        self.conv_blk = ffn(hidden_dim)
        self.ln_2 = nn.LayerNorm(hidden_dim)
        self.skip_scale2 = nn.Parameter(torch.ones(hidden_dim))

    def _fused_ok(self, x, width=None, height=None):
        """Whole-block HIP path: inference on the HIP backend for the kernel's shape range (`width`, `height` of the map)."""
        ops = _OpsBackend.impl
        ss = self.self_attention
        if not (hasattr(ops, "lfss_block_forward") and x.is_cuda and x.dtype == torch.float32):
            return False
        if _needs_grad(self, x):
            return False
        if ss.dropout is not None or ss.in_proj.bias is not None or ss.out_proj.bias is not None:
            return False
        return ops.lfss_block_supported(ss.d_model, ss.d_inner, ss.d_state, ss.dt_rank,
                                        self.conv_blk.conv1.out_channels, width, height)

    def _nchw_train_ok(self, x):
        """Training on the HIP backend: the block on NCHW planes, every operator but the gates / skips an autograd
        Function over HIP kernels (LayerNorm2d, 1x1 convolutions, depth-wise conv, the SS2D core) - no token <-> map
        permutes.  Same shape range as the fused inference path."""
        ops = _OpsBackend.impl
        ss = self.self_attention
        if not (torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
            return False
        if not all(hasattr(ops, n) for n in ("layernorm2d_train", "conv2d_train", "dwconv3x3_train", "ss2d_core")):
            return False
        if ss.dropout is not None or ss.in_proj.bias is not None or ss.out_proj.bias is not None:
            return False
        if not (isinstance(self.ln_1, nn.LayerNorm) and isinstance(self.ln_2, nn.LayerNorm)):
            return False
        return (ops.lfss_block_supported(ss.d_model, ss.d_inner, ss.d_state, ss.dt_rank, self.conv_blk.conv1.out_channels)
                and ops.ss2d_core_bwd_supported(ss.d_inner, ss.d_state, ss.dt_rank))

    def forward_nchw_train(self, x):
        """(B, C, H, W) -> (B, C, H, W); the same arithmetic as forward() on the (B, HW, C) view (:520-528)."""
        ops = _OpsBackend.impl
        ss = self.self_attention
        C, D = ss.d_model, ss.d_inner
        a = ops.layernorm2d_train(x, self.ln_1.weight, self.ln_1.bias, self.ln_1.eps)
        w_in = ss.in_proj.weight.view(2 * D, C, 1, 1)
        xs = ops.conv2d_train(a, w_in[:D])                                      # nn.Linear = 1x1 convolution
        z = ops.conv2d_train(a, w_in[D:])
        xs = F.silu(ops.dwconv3x3_train(xs, ss.conv2d.weight, ss.conv2d.bias))
        y = ops.ss2d_core(xs, ss.x_proj_weight, ss.dt_projs_weight, ss.dt_projs_bias, ss.A_logs, ss.Ds, merged=True)
        y = ops.layernorm2d_train(y.view_as(xs), ss.out_norm.weight, ss.out_norm.bias, ss.out_norm.eps)
        fused = hasattr(ops, "gate_act") and hasattr(ops, "scale_add")           # gates / skips: one HIP launch each way
        gated = ops.gate_act(z, y, "silu") if fused and ops.gate_supported(z, y) else y * F.silu(z)
        o = ops.conv2d_train(gated, ss.out_proj.weight.view(C, D, 1, 1))
        if not fused:
            t = x * self.skip_scale.view(1, -1, 1, 1) + o
            u = ops.layernorm2d_train(t, self.ln_2.weight, self.ln_2.bias, self.ln_2.eps)
            return t * self.skip_scale2.view(1, -1, 1, 1) + self.conv_blk(u)
        t = ops.scale_add(x, self.skip_scale, o)
        u = ops.layernorm2d_train(t, self.ln_2.weight, self.ln_2.bias, self.ln_2.eps)
        return ops.scale_add(t, self.skip_scale2, self.conv_blk(u))

    def forward(self, input, x_size):
        B, L, C = input.shape
        if self._fused_ok(input, x_size[1], x_size[0]):
            return _OpsBackend.impl.lfss_block_forward(input, x_size, self)
        tok = input.view(B, x_size[0], x_size[1], C)
        tok = tok * self.skip_scale + self.drop_path(self.self_attention(_ln_tok(self.ln_1, tok)))
        mix = self.conv_blk(_ln_tok(self.ln_2, tok).permute(0, 3, 1, 2).contiguous()).permute(0, 2, 3, 1)
        tok = tok * self.skip_scale2 + mix
        return tok.reshape(B, L, C)


# ================================================================================================
# High-frequency branch: HFEBlock family (HIP kernels in inference, PyTorch autograd in training)
# ================================================================================================
class LayerNorm2d(nn.Module):
    """Per-pixel LayerNorm over channels of an NCHW map, eps 1e-6 (reference :532-569)."""

    def __init__(self, channels, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(channels))
        self.bias = nn.Parameter(torch.zeros(channels))
        self.eps = eps

    def forward(self, x):
        ops = _OpsBackend.impl
        if hasattr(ops, "layernorm2d") and x.is_cuda and x.dtype == torch.float32 and x.shape[1] in (8, 16, 32):
            if not _needs_grad(self, x):
                return ops.layernorm2d(x, self.weight, self.bias, self.eps)
            return ops.layernorm2d_train(x, self.weight, self.bias, self.eps)
        mu = x.mean(1, keepdim=True)
        var = (x - mu).pow(2).mean(1, keepdim=True)
        y = (x - mu) / (var + self.eps).sqrt()
        return self.weight.view(1, -1, 1, 1) * y + self.bias.view(1, -1, 1, 1)


def _gram_ok(a, b, differentiable=True):
    """The HIP Gram kernel serves (a, b); `differentiable=False`: the caller takes nothing differentiable from it (the
    channel matching keeps indices only), so tensors under autograd qualify too."""
    ops = _OpsBackend.impl
    return (hasattr(ops, "gram") and a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32
            and a.shape == b.shape and a.shape[1] <= 32
            and not (differentiable and torch.is_grad_enabled() and (a.requires_grad or b.requires_grad)))


def nearest_candidate_index(maps, candidates, num_matches):
    """Channel matching (reference :618-666): for every channel of `maps` (B, C, HW) the index of its L2-nearest
    channel of `candidates`; keeps the `num_matches` channels whose nearest distance is smallest (original
    channel order).  -> (B, num_matches) int64 channel indices into `candidates`."""
    ops = _OpsBackend.impl
    if _gram_ok(maps, candidates, differentiable=False):
        # d^2 = |x|^2 + |y|^2 - 2 x.y, the same expansion torch.cdist uses for C > 25 (mm mode).  Indices only leave this
        # function (topk indices are not differentiable in the reference either): training takes the same kernel
        G, nx, ny = ops.gram(maps.detach(), candidates.detach())
        if hasattr(ops, "match_index") and (num_matches is None or num_matches == -1 or num_matches >= maps.size(1)):
            return ops.match_index(G, nx, ny)                       # every channel kept: one argmin kernel
        dist = (nx.unsqueeze(2) + ny.unsqueeze(1) - 2.0 * G).clamp_min(1e-30).sqrt()
    else:
        dist = torch.cdist(maps, candidates)                        # (B, C, C)
    best_val, best_idx = dist.topk(k=1, largest=False)
    best_val, best_idx = best_val.squeeze(-1), best_idx.squeeze(-1)
    if num_matches is None or num_matches == -1:
        num_matches = maps.size(1)
    if num_matches < maps.size(1):
        rank = best_val.argsort(dim=1).argsort(dim=1)               # rank of each channel's distance
        best_idx = best_idx.masked_select(rank < num_matches).reshape(maps.size(0), num_matches)
    return best_idx


def nearest_candidate_maps(maps, candidates, num_matches):
    """The matched candidate maps themselves: gather(candidates, nearest_candidate_index) -> (B, num_matches, HW)."""
    best_idx = nearest_candidate_index(maps, candidates, num_matches)
    gather_idx = best_idx.long().unsqueeze(-1).expand(-1, -1, candidates.size(2))
    return torch.gather(candidates, 1, gather_idx)


class Matching(nn.Module):
    def __init__(self, dim=32, match_factor=1):
        super().__init__()
        self.num_matching = int(dim / match_factor)

    def forward(self, x, perception):
        b, c, h, w = x.shape
        matched = nearest_candidate_maps(x.flatten(2), perception.flatten(2), self.num_matching)
        return matched.reshape(b, self.num_matching, h, w)


class PAConv(nn.Module):
    """Pixel-attention conv: k4(k3(x) * sigmoid(k2(x))) (reference :683-700).  `x` may be given as the pair
    (x, x2[, x2_index]) standing for cat([x, gather(x2, x2_index)], 1)."""

    def __init__(self, nf, k_size=3):
        super().__init__()
        self.k2 = nn.Conv2d(nf, nf, 1)
        self.sigmoid = nn.Sigmoid()
        self.k3 = nn.Conv2d(nf, nf, kernel_size=k_size, padding=(k_size - 1) // 2, bias=False)
        self.k4 = nn.Conv2d(nf, nf // 2, kernel_size=k_size, padding=(k_size - 1) // 2, bias=False)

    def forward(self, x, x2=None, x2_index=None):
        ops = _OpsBackend.impl
        if (hasattr(ops, "conv2d_gated") and self.k3.bias is None and tuple(self.k3.weight.shape[2:]) == (3, 3)
                and ops.conv2d_supported(x, self.k3.weight, x2)
                and not (torch.is_grad_enabled() and any(t is not None and t.requires_grad
                                                         for t in (x, x2, self.k3.weight, self.k2.weight)))):
            # k3(x) * sigmoid(k2(x)): the 1x1 rides on the 3x3's centre tap, the gate never exists as a tensor
            return _conv(self.k4, ops.conv2d_gated(x, self.k3.weight, self.k2.weight, self.k2.bias, x2, x2_index))
        xin = _cat_gathered(x, x2, x2_index)            # once for both convolutions (and one scatter in the backward)
        gate = _conv(self.k2, xin)
        return _conv(self.k4, _conv(self.k3, xin, gate=gate))


class Matching_transformation(nn.Module):
    def __init__(self, dim=32, match_factor=1, ffn_expansion_factor=1, bias=True):
        super().__init__()
        self.num_matching = int(dim / match_factor)
        self.channel = dim
        self.matching = Matching(dim=dim, match_factor=match_factor)
        self.paconv = PAConv(dim * 2)

    def forward(self, x, perception):
        # reference :713: paconv(cat([x, matching(x, perception)], 1)); the matched maps are a channel gather of
        # `perception`, handed to the convolutions as (perception, index)
        idx = nearest_candidate_index(x.flatten(2), perception.flatten(2), self.num_matching)
        return self.paconv(x, perception, idx)


class FeedForward(nn.Module):
    """HFE feed-forward (reference :721-751)."""

    def __init__(self, dim=32, match_factor=4, ffn_expansion_factor=1, bias=True, ffn_matching=True):
        super().__init__()
        self.num_matching = int(dim / match_factor)
        self.channel = dim
        self.matching = ffn_matching
        hidden = int(dim * ffn_expansion_factor)
        self.project_in = nn.Sequential(
            nn.Conv2d(dim, hidden, 1, bias=bias),
            nn.Conv2d(hidden, dim, kernel_size=3, stride=1, padding=1, groups=dim, bias=bias))
        if self.matching is True:
            self.matching_transformation = Matching_transformation(
                dim=dim, match_factor=match_factor, ffn_expansion_factor=ffn_expansion_factor, bias=bias)
        self.project_out = nn.Sequential(
            nn.Conv2d(dim, hidden, kernel_size=3, stride=1, padding=1, groups=dim, bias=bias),
            nn.GELU(),
            nn.Conv2d(hidden, dim, 1, bias=bias))

    def forward(self, x, perception, residual=None, norm=None):
        """`norm`: the block's norm2, applied to `x` here (so that it can ride in project_in's 1x1 kernel)"""
        y = _dwconv(self.project_in[1], _ln_conv(norm, self.project_in[0], x))
        if perception is not None:
            y = self.matching_transformation(y, perception)
        # project_out = [depth-wise 3x3, GELU, 1x1]: the GELU rides in the depth-wise kernel
        return _conv(self.project_out[2], _dwconv(self.project_out[0], y, act="gelu"), residual=residual)


class CMTAttention(nn.Module):
    """Transposed (channel) attention with channel-matched queries (reference :756-798)."""

    def __init__(self, dim, num_heads, match_factor=4, ffn_expansion_factor=1, scale_factor=8, bias=True,
                 attention_matching=True):
        super().__init__()
        self.num_heads = num_heads
        self.temperature = nn.Parameter(torch.ones(num_heads, 1, 1))
        self.qkv = nn.Conv2d(dim, dim * 3, kernel_size=1, bias=bias)
        self.qkv_dwconv = nn.Conv2d(dim * 3, dim * 3, kernel_size=3, stride=1, padding=1, groups=dim * 3,
                                    bias=bias)
        self.project_out = nn.Conv2d(dim, dim, kernel_size=1, bias=bias)
        self.matching = attention_matching
        if self.matching is True:
            self.matching_transformation = Matching_transformation(
                dim=dim, match_factor=match_factor, ffn_expansion_factor=ffn_expansion_factor, bias=bias)

    def qkv_of(self, x, norm=None):
        """The part of `forward` that needs `x` alone (UNet.forward runs it ahead of time on a side stream).  `norm`: the
        block's norm1, applied to `x` here (it rides in the qkv 1x1 kernel)."""
        return _dwconv(self.qkv_dwconv, _ln_conv(norm, self.qkv, x))

    def forward(self, x, perception, residual=None, qkv=None, norm=None):
        q, k, v = (self.qkv_of(x, norm) if qkv is None else qkv).chunk(3, dim=1)
        x = v                                              # from here on only shape / device / dtype of `x` matter
        b, c, h, w = x.shape
        if self.matching is True:
            q = self.matching_transformation(q, perception)
        heads = self.num_heads
        q = q.reshape(b * heads, c // heads, h * w)
        k = k.reshape(b * heads, c // heads, h * w)
        v = v.reshape(b, heads, c // heads, h * w)
        ops = _OpsBackend.impl
        train_gram = (hasattr(ops, "gram_train") and _gram_ok(q, k, differentiable=False)
                      and torch.is_grad_enabled() and (q.requires_grad or k.requires_grad))
        if train_gram or _gram_ok(q, k):
            # normalize(q) @ normalize(k)^T == (q @ k^T) / (max(|q|, eps) max(|k|, eps)): one pass over q, k
            G, nq, nk = ops.gram_train(q, k) if train_gram else ops.gram(q.contiguous(), k.contiguous())
            if (hasattr(ops, "attn_fold") and c <= 64 and ops.conv2d_supported(x, self.project_out.weight)
                    and not (torch.is_grad_enabled() and any(t.requires_grad for t in self.parameters()))):
                # project_out(softmax(...) @ v) = (W_po @ blockdiag(attn)) @ v: a tiny kernel folds the (c/heads)^2
                # attention into the 1x1 weight, the 1x1 convolution kernel applies it (+ bias, + residual)
                wf = ops.attn_fold(G, nq, nk, self.temperature.reshape(heads), self.project_out.weight, b, heads)
                vv = v.reshape(b, c, h, w)
                outs = [ops.conv2d(vv[i:i + 1], wf[i].view(c, c, 1, 1), self.project_out.bias,
                                   residual=None if residual is None else residual[i:i + 1], dynamic_weight=True)
                        for i in range(b)]
                return outs[0] if b == 1 else torch.cat(outs, 0)
            # clamp BEFORE the root: sqrt'(0) is infinite, and clamp's zero gradient times it is NaN for an all-zero q / k row
            # (F.normalize, the reference's spelling, gives a finite zero gradient there) - ADVICE r4
            scale = nq.clamp_min(1e-24).sqrt().unsqueeze(2) * nk.clamp_min(1e-24).sqrt().unsqueeze(1)
            attn = (G / scale).reshape(b, heads, c // heads, c // heads)
        else:
            qn = F.normalize(q.reshape(b, heads, c // heads, h * w), dim=-1)
            kn = F.normalize(k.reshape(b, heads, c // heads, h * w), dim=-1)
            attn = qn @ kn.transpose(-2, -1)
        attn = (attn * self.temperature).softmax(dim=-1)
        return _conv(self.project_out, (attn @ v).reshape(b, c, h, w), residual=residual)


class HFEBlock(nn.Module):
    """High-frequency enhancement block (reference :822-854); `perception` is the low-freq map."""

    def __init__(self, dim=48, num_heads=1, match_factor=4, ffn_expansion_factor=1, bias=True,
                 attention_matching=True, ffn_matching=True, ffn_restormer=False):
        super().__init__()
        if ffn_restormer:
            raise NotImplementedError("ffn_restormer=True is never instantiated by the reference UNet")
        self.dim = dim
        self.norm1 = LayerNorm2d(dim)
        self.attn = CMTAttention(dim=dim, num_heads=num_heads, match_factor=match_factor,
                                 ffn_expansion_factor=ffn_expansion_factor, bias=bias,
                                 attention_matching=attention_matching)
        self.norm2 = LayerNorm2d(dim)
        self.ffn_restormer = ffn_restormer
        self.ffn = FeedForward(dim=dim, match_factor=match_factor, ffn_expansion_factor=ffn_expansion_factor,
                               bias=bias, ffn_matching=ffn_matching)
        self.LayerNorm = LayerNorm2d(dim)

    def qkv_of(self, x):
        """dwconv(qkv(norm1(x))): everything of the block that does not need `perception`."""
        return self.attn.qkv_of(x, self.norm1)

    def forward(self, x, perception, qkv=None):
        p = self.LayerNorm(perception)
        x = self.attn(x if qkv is None else None, p, residual=x, qkv=qkv, norm=self.norm1)   # x + attn(norm1(x)): the add rides in the 1x1 epilogue
        return self.ffn(x, p, residual=x, norm=self.norm2)


class SKFF(nn.Module):
    """Selective-kernel fusion of the three detail sub-bands (reference :923-959)."""

    def __init__(self, in_channels, height=3, reduction=8, bias=False):
        super().__init__()
        self.height = height
        d = max(int(in_channels / reduction), 4)
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.conv_du = nn.Sequential(nn.Conv2d(in_channels, d, 1, padding=0, bias=bias), nn.PReLU())
        self.fcs = nn.ModuleList([nn.Conv2d(d, in_channels, kernel_size=1, stride=1, bias=bias)
                                  for _ in range(height)])
        self.softmax = nn.Softmax(dim=1)

    def forward(self, inp_feats):
        b, c = inp_feats[0].shape[:2]
        ops = _OpsBackend.impl
        x0 = inp_feats[0]
        if (hasattr(ops, "skff") and self.height == 3 and len(inp_feats) == 3 and x0.is_cuda and x0.dtype == torch.float32
                and c <= 64 and self.conv_du[0].weight.shape[0] <= 16 and self.conv_du[1].weight.numel() == 1
                and not (torch.is_grad_enabled() and (any(t.requires_grad for t in inp_feats)
                                                      or any(p.requires_grad for p in self.parameters())))):
            w_fc = torch.stack([fc.weight.reshape(c, -1) for fc in self.fcs], 0)      # (3, C, d)
            return ops.skff(inp_feats[0], inp_feats[1], inp_feats[2], self.conv_du[0].weight, self.conv_du[1].weight, w_fc)
        stack = torch.stack(list(inp_feats), dim=1)                    # (B, height, C, H, W)
        squeeze = self.conv_du(self.avg_pool(stack.sum(dim=1)))
        weights = torch.cat([fc(squeeze) for fc in self.fcs], dim=1).view(b, self.height, c, 1, 1)
        return (stack * self.softmax(weights)).sum(dim=1)


# ================================================================================================
# Wavelet U-Net
# ================================================================================================
def _run_lfss_stack(blocks, x):
    """(B, C, H, W) -> (B, C, H, W) through a stack of LFSSBlocks.  On the fused HIP path the first block
    reads NCHW and the last writes NCHW, so the `b c h w <-> b (h w) c` rearranges of the reference
    (:976-979, :998-1001) never materialise; otherwise the reference's token round trip."""
    h, w = x.shape[2:]
    blocks = list(blocks)
    if blocks and all(blk._fused_ok(x, w, h) for blk in blocks):
        ops = _OpsBackend.impl
        t = x
        for i, blk in enumerate(blocks):
            t = ops.lfss_block_forward(t, (h, w), blk, tok_nchw=(i == 0), out_nchw=(i == len(blocks) - 1))
        return t
    if blocks and all(blk._nchw_train_ok(x) for blk in blocks):
        for blk in blocks:
            x = blk.forward_nchw_train(x)
        return x
    t = _tokens(x)
    for blk in blocks:
        t = blk(t, [h, w])
    return _maps(t, h, w)


def _tokens(x):        # (B, C, H, W) -> (B, HW, C)
    return x.flatten(2).transpose(1, 2).contiguous()


def _maps(t, h, w):    # (B, HW, C) -> (B, C, H, W)
    return t.transpose(1, 2).reshape(t.shape[0], t.shape[2], h, w).contiguous()


class DownFRG(nn.Module):
    """DWT -> low-freq LFSS stack + high-freq SKFF/HFE stack (reference :962-985)."""

    early_qkv = True       # side-stream order only: the first HFEBlock's qkv head issued before the LFSS stack's result is waited for

    def __init__(self, dim, n_l_blocks=1, n_h_blocks=1, expand=2):
        super().__init__()
        self.dwt = DWT()
        self.l_conv = nn.Conv2d(dim * 2, dim, 3, 1, 1)
        self.l_blk = nn.Sequential(*[LFSSBlock(dim, expand=expand) for _ in range(n_l_blocks)])
        self.h_fusion = SKFF(dim, height=3, reduction=8)
        self.h_blk = nn.Sequential(*[HFEBlock(dim, match_factor=1, ffn_expansion_factor=1)
                                     for _ in range(n_h_blocks)])

    def forward(self, x, x_d, side=None, x_d_ready=None):
        """`side`: a second CUDA stream for the high-frequency branch (UNet.forward, inference only).  The branch needs
        nothing but this level's sub-bands and `low`, and nothing needs it before the matching up group: issued on its
        own stream it runs under the deeper levels' kernels (whose grids - 130 k positions at level 3 - leave compute
        units idle).  Returns (low, high); with `side`, `high` is complete on `side` (the caller joins)."""
        ll, hl, lh, hh = self.dwt(x)
        if side is None:
            low = _run_lfss_stack(self.l_blk, _conv(self.l_conv, ll, x_d))
            high = self.h_fusion([hl, lh, hh])
            for blk in self.h_blk:
                high = blk(high, low)
            return low, high
        main = torch.cuda.current_stream(x.device)
        side.wait_stream(main)                         # the sub-bands are ready
        with torch.cuda.stream(side):
            high = self.h_fusion([hl, lh, hh])
            # the part of the first HFEBlock that needs `high` alone (norm1 -> qkv 1x1 -> depth-wise 3x3) runs under the main
            # stream's LFSS stack too, not behind it (the up groups do the same with their branch, UNet.forward)
            qkv0 = self.h_blk[0].qkv_of(high) if (len(self.h_blk) and self.early_qkv) else None
        if x_d_ready is not None:                      # x_d was computed on `side` (UNet.forward)
            main.wait_event(x_d_ready)
            x_d.record_stream(main)
        low = _run_lfss_stack(self.l_blk, _conv(self.l_conv, ll, x_d))
        side.wait_stream(main)                         # low is ready
        with torch.cuda.stream(side):
            for i, blk in enumerate(self.h_blk):
                high = blk(high, low, qkv=qkv0 if i == 0 else None)
        for t in (hl, lh, hh, low):                    # allocated on `main`, read on `side`
            t.record_stream(side)
        return low, high


class upFRG(nn.Module):
    """LFSS stack + HFE stack -> IWT (reference :987-1008)."""

    def __init__(self, dim, n_l_blocks=1, n_h_blocks=1, expand=2):
        super().__init__()
        self.iwt = IWT()
        self.l_blk = nn.Sequential(*[LFSSBlock(dim, expand=expand) for _ in range(n_l_blocks)])
        self.h_out_conv = nn.Conv2d(dim, dim * 3, 3, 1, 1)
        self.h_blk = nn.Sequential(*[HFEBlock(dim, match_factor=1, ffn_expansion_factor=1)
                                     for _ in range(n_h_blocks)])

    def forward(self, x_l, x_h, join=None):
        """`join`: makes a side-stream `x_h` (DownFRG.forward) an input of the current stream - called only after this
        group's LFSS stack, which does not need it, has been issued."""
        low = _run_lfss_stack(self.l_blk, x_l)
        qkv0 = None
        if join is not None:
            x_h, qkv0 = join(x_h)
        for i, blk in enumerate(self.h_blk):
            x_h = blk(x_h, low, qkv=qkv0 if i == 0 else None)
        # reference: iwt(cat([x_l, h_out_conv(x_h)], 1)); the pair form skips the concatenation
        return self.iwt(low, _conv(self.h_out_conv, x_h))


_SIDE_STREAMS = {}


def _side_streams(x, n):
    """The side streams of the inference forward, per (device, the stream the forward is issued on): forwards in flight
    on different main streams (multi-stream serving, bench.py's concurrent leg) must not share side streams, or each
    one's join would also wait for the others' high-frequency work.  Under HIP-graph capture no stream may be created
    (the capture would be invalidated): a per-device set reserved for captures is made together with the first ordinary
    one.  ((None,) * n off a GPU.)"""
    if not x.is_cuda:
        return (None,) * n
    dev = x.device.index if x.device.index is not None else torch.cuda.current_device()
    capturing = torch.cuda.is_current_stream_capturing()
    key = (dev, "capture" if capturing else torch.cuda.current_stream(x.device).cuda_stream)
    sts = _SIDE_STREAMS.get(key)
    if sts is None or len(sts) < n:
        if capturing:
            raise RuntimeError("WaveMamba: run one eager forward on this device before capturing it into a graph "
                               "(the side streams of a capture are created outside of it)")
        sts = _SIDE_STREAMS[key] = tuple(torch.cuda.Stream(device=x.device) for _ in range(n))
        cap = _SIDE_STREAMS.get((dev, "capture"))
        if cap is None or len(cap) < n:
            _SIDE_STREAMS[(dev, "capture")] = tuple(torch.cuda.Stream(device=x.device) for _ in range(n))
    return sts[:n]


class UNet(nn.Module):
    """Three-level wavelet U-Net (reference :1011-1063)."""

    # inference: the down path's high-frequency branches on side streams (DownFRG.forward); WM_TWO_STREAMS=0 or
    # `net.restoration_network.two_streams = False` for the single-stream order
    two_streams = os.environ.get("WM_TWO_STREAMS", "1") != "0"

    def __init__(self, in_chn=3, wf=48, n_l_blocks=[1, 1, 2], n_h_blocks=[1, 1, 1], ffn_scale=2):
        super().__init__()
        self.ps_down1 = nn.Sequential(nn.PixelUnshuffle(2), nn.Conv2d(4 * in_chn, wf, 1, 1, 0))
        self.ps_down2 = nn.Sequential(nn.PixelUnshuffle(4), nn.Conv2d(16 * in_chn, wf, 1, 1, 0))
        self.ps_down3 = nn.Sequential(nn.PixelUnshuffle(8), nn.Conv2d(64 * in_chn, wf, 1, 1, 0))
        self.conv_01 = nn.Conv2d(in_chn, wf, 3, 1, 1)
        self.down_group1 = DownFRG(wf, n_l_blocks[0], n_h_blocks[0], expand=ffn_scale)
        self.down_group2 = DownFRG(wf, n_l_blocks[1], n_h_blocks[1], expand=ffn_scale)
        self.down_group3 = DownFRG(wf, n_l_blocks[2], n_h_blocks[2], expand=ffn_scale)
        self.up_group3 = upFRG(wf, n_l_blocks[2], n_h_blocks[2], expand=ffn_scale)
        self.up_group2 = upFRG(wf, n_l_blocks[1], n_h_blocks[1], expand=ffn_scale)
        self.up_group1 = upFRG(wf, n_l_blocks[0], n_h_blocks[0], expand=ffn_scale)
        self.last = nn.Conv2d(wf, in_chn, kernel_size=3, stride=1, padding=1, bias=True)

    def forward(self, x):
        img = x
        # one side stream per level: level 1's branch (the largest) is not needed before the last up group.
        # (Rounds 4-5: the multi-stream order gave run-to-run different results with bf16 planes.  Root cause, DESIGN.md 7 /
        # profiles/r05/: not a missing dependency but ONE instruction form the SLP vectoriser had produced in dwconv3x3<bf16> -
        # `v_pk_fma_f32 v, s[..], v, v op_sel:[0,0,1]`: a packed-fp32 op with a scalar source and a half-swapped VGPR source
        # returns zero for that half in lanes 48..63 while LDS-fed MFMAs of another kernel share the SIMD (standalone reproducer:
        # tools/ubench_pk_coexec.hip).  The library is built without that form now and tools/lint_packed_f32.py keeps it out.)
        sides = (_side_streams(x, 3) if self.two_streams and not _needs_grad(self, x) else (None, None, None))
        pss = (self.ps_down1, self.ps_down2, self.ps_down3)
        ops = _OpsBackend.impl
        fused_ps = all(hasattr(ops, "patchify_conv") and ops.patchify_conv_supported(img, ps[1].weight, ps[0].downscale_factor)
                       for ps in pss)
        if sides[0] is None or fused_ps:
            # (the fused patch embeddings are 0.18 ms per UHD image together: issued on the main stream)
            d, d_ready = [_ps_conv(ps, img) for ps in pss], (None, None, None)
        else:                                          # the pixel-unshuffled inputs of the three l_convs: off the main chain too
            main = torch.cuda.current_stream(x.device)
            d, d_ready = [], []
            for ps, side in zip(pss, sides):
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    d.append(_ps_conv(ps, img))
                    d_ready.append(side.record_event())
            img.record_stream(sides[0]); img.record_stream(sides[1]); img.record_stream(sides[2])
        low, high1 = self.down_group1(_conv(self.conv_01, img), d[0], sides[0], d_ready[0])
        low, high2 = self.down_group2(low, d[1], sides[1], d_ready[1])
        low, high3 = self.down_group3(low, d[2], sides[2], d_ready[2])

        def joiner(side, up_group, high):              # the branch becomes an input of the main stream, as late as possible
            if side is None:
                return None
            with torch.cuda.stream(side):              # ... after the part of the up group's first HFEBlock that needs only it
                qkv = up_group.h_blk[0].qkv_of(high) if len(up_group.h_blk) else None

            def join(high):
                main = torch.cuda.current_stream(x.device)
                main.wait_stream(side)
                for t in (high, qkv):                  # allocated on `side`, read on `main`
                    if t is not None:
                        t.record_stream(main)
                return high, qkv
            return join
        low = self.up_group3(low, high3, joiner(sides[2], self.up_group3, high3))
        low = self.up_group2(low, high2, joiner(sides[1], self.up_group2, high2))
        low = self.up_group1(low, high1, joiner(sides[0], self.up_group1, high1))
        return _conv(self.last, low, residual=img)


@ARCH_REGISTRY.register()
class WaveMamba(nn.Module):
    """Registry entry (reference :1066-1176).  Input NCHW in [0, 1], H and W multiples of 8."""

    scale_factor = 1          # the reference's test_tile reads an undefined self.scale_factor (:1099)

    def __init__(self, *, in_chn, wf, n_l_blocks=[1, 1, 2], n_h_blocks=[1, 1, 1], ffn_scale=2.0,
                 **ignore_kwargs):
        super().__init__()
        self.restoration_network = UNet(in_chn=in_chn, wf=wf, n_l_blocks=n_l_blocks,
                                        n_h_blocks=n_h_blocks, ffn_scale=ffn_scale)

    # The inference convolutions keep prepared (bf16-split) copies of their weights, validated by the parameters'
    # version counters.  Writes through `.data` (EMA updates, weight surgery) do not bump those: every switch of mode
    # and every device / dtype move drops the copies, so `net.eval()` after such an update is enough.
    def train(self, mode=True):
        _hip_ops.conv2d_cache_clear()
        return super().train(mode)

    def _apply(self, fn, *args, **kwargs):
        _hip_ops.conv2d_cache_clear()
        return super()._apply(fn, *args, **kwargs)

    def print_network(self, model):
        print(model)
        print("The number of parameters: {}".format(sum(p.numel() for p in model.parameters())))

    def encode_and_decode(self, input, current_iter=None):
        return self.restoration_network(input)

    def check_image_size(self, x, window_size=8):
        _, _, h, w = x.size()
        pad_h = (window_size - h % window_size) % window_size
        pad_w = (window_size - w % window_size) % window_size
        return F.pad(x, (0, pad_w, 0, pad_h), "reflect")

    @torch.no_grad()
    def test(self, input):
        return self.encode_and_decode(input)

    @torch.no_grad()
    def test_tile(self, input, tile_size=240, tile_pad=16):
        """Tile-by-tile inference with padded borders (reference :1091-1151, scale factor 1)."""
        batch, channel, height, width = input.shape
        s = self.scale_factor
        output = input.new_zeros((batch, channel, height * s, width * s))
        for y0 in range(0, height, tile_size):
            for x0 in range(0, width, tile_size):
                y1, x1 = min(y0 + tile_size, height), min(x0 + tile_size, width)
                py0, px0 = max(y0 - tile_pad, 0), max(x0 - tile_pad, 0)
                py1, px1 = min(y1 + tile_pad, height), min(x1 + tile_pad, width)
                tile = self.test(input[:, :, py0:py1, px0:px1])
                oy, ox = (y0 - py0) * s, (x0 - px0) * s
                output[:, :, y0 * s:y1 * s, x0 * s:x1 * s] = \
                    tile[:, :, oy:oy + (y1 - y0) * s, ox:ox + (x1 - x0) * s]
        return output

    def forward(self, input):
        return self.encode_and_decode(input)
