"""Minimal trainer counterpart for the hot path's caller (reference basicsr/models/femasr_model.py).

Reproduces exactly what `FeMaSRModel.optimize_parameters` (:157-185) does to the network:
    loss = L1(output, gt) + 0.1 * L1(stack(rfft2(output).real/imag), stack(rfft2(gt).real/imag))
(`self.l1` is a bare nn.L1Loss :30,:171; FFTLoss losses.py:306-313 with fft_opt.loss_weight 0.1,
train_wavemamba_uhdll.yml:102-104), AdamW(lr 5e-4, weight_decay 1e-3, betas (0.9, 0.99)) (yml:75-79),
DistributedDataParallel wrap with one gradient all-reduce per step (base_model.py:111-114; backend
'nccl' == RCCL on ROCm), and the per-iteration loss reduce to rank 0 (base_model.py:376-401).
"""
import torch
import torch.distributed as dist
import torch.nn.functional as F


def fft_l1(pred, target):
    """FFTLoss (losses.py:306-313): L1 between the stacked real/imag parts of rfft2."""
    pf, tf = torch.fft.rfft2(pred), torch.fft.rfft2(target)
    return F.l1_loss(torch.stack([pf.real, pf.imag], dim=-1), torch.stack([tf.real, tf.imag], dim=-1))


def losses(output, gt, fft_weight=0.1):
    return F.l1_loss(output, gt), fft_weight * fft_l1(output, gt)


def make_optimizer(net, lr=5e-4, weight_decay=1e-3, betas=(0.9, 0.99)):
    """AdamW as the reference configures it (train_wavemamba_uhdll.yml:75-79).  On a GPU the 591 small tensors are
    updated by the fused multi-tensor implementation (one launch per ~hundred tensors instead of ~10 per tensor
    group); same arithmetic."""
    params = [p for p in net.parameters() if p.requires_grad]
    fused = bool(params) and all(p.is_cuda for p in params)
    return torch.optim.AdamW(params, lr=lr, weight_decay=weight_decay, betas=betas, fused=fused)


def wrap_ddp(net, device=None, find_unused_parameters=False):
    """DDP wrap like base_model.py:111-114.  find_unused_parameters defaults to False: every one of
    the 591 parameter tensors receives a gradient, so the graph walk is wasted work."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return net
    ids = [device.index] if device is not None and device.type == "cuda" else None
    return torch.nn.parallel.DistributedDataParallel(net, device_ids=ids,
                                                     find_unused_parameters=find_unused_parameters)


def train_step(net, optimizer, lq, gt):
    """One optimize_parameters(): zero_grad, forward, L1 + 0.1*FFT, backward (DDP all-reduce), step."""
    optimizer.zero_grad(set_to_none=True)
    out = net(lq)
    l_pix, l_freq = losses(out, gt)
    (l_pix + l_freq).mean().backward()
    optimizer.step()
    return reduce_loss_dict({"l_pix": l_pix.detach(), "l_freq": l_freq.detach()})


def reduce_loss_dict(loss_dict):
    """base_model.py:376-401: sum-reduce to rank 0 then divide by world size."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return {k: float(v) for k, v in loss_dict.items()}
    keys = sorted(loss_dict)
    vec = torch.stack([loss_dict[k].float() for k in keys])
    dist.reduce(vec, dst=0)
    if dist.get_rank() == 0:
        vec /= dist.get_world_size()
    return {k: float(v) for k, v in zip(keys, vec)}
