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


def _on_hip(a, b):
    return a.is_cuda and b.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.shape == b.shape


def l1(pred, target):
    """nn.L1Loss() (femasr_model.py:30, :171).  GPU fp32 tensors: the HIP reduction (ops.l1_mean) - no ATen semaphore memset in the
    step, so the step can be captured into a HIP graph and the REPORTED loss of a replay stays right."""
    if _on_hip(pred, target):
        from . import ops
        return ops.l1_mean(pred, target)
    return F.l1_loss(pred, target)


def fft_l1(pred, target):
    """FFTLoss (losses.py:306-313): L1 between the stacked real/imag parts of rfft2.  On the GPU the same mean is taken over the
    interleaved (re, im) floats of the complex tensors (`view_as_real`: the stacked copies are never made)."""
    pf, tf = torch.fft.rfft2(pred), torch.fft.rfft2(target)
    if pf.is_cuda and pf.dtype == torch.complex64 and tf.dtype == torch.complex64:
        return l1(torch.view_as_real(pf), torch.view_as_real(tf))
    return F.l1_loss(torch.stack([pf.real, pf.imag], dim=-1), torch.stack([tf.real, tf.imag], dim=-1))


def losses(output, gt, fft_weight=0.1):
    return l1(output, gt), fft_weight * fft_l1(output, gt)


def make_optimizer(net, lr=5e-4, weight_decay=1e-3, betas=(0.9, 0.99), capturable=False):
    """AdamW as the reference configures it (train_wavemamba_uhdll.yml:75-79).  On a GPU the 591 small tensors are
    updated by the fused multi-tensor implementation (one launch per ~hundred tensors instead of ~10 per tensor
    group); same arithmetic.  capturable: step counters AND the learning rate on the device, for GraphedTrainStep - a
    Python-float lr would be baked into the captured AdamW launch as a kernel scalar and a scheduler (the reference's recipe
    uses CosineAnnealingRestartCyclicLR, train_wavemamba_uhdll.yml:86-90) would change nothing in the replays; torch's
    schedulers `fill_()` a tensor lr in place, which the replayed kernel reads."""
    params = [p for p in net.parameters() if p.requires_grad]
    fused = bool(params) and all(p.is_cuda for p in params)
    kw = {"capturable": True} if capturable else {}
    if capturable and fused:
        lr = torch.tensor(float(lr), dtype=torch.float32, device=params[0].device)
    return torch.optim.AdamW(params, lr=lr, weight_decay=weight_decay, betas=betas, fused=fused, **kw)


def wrap_ddp(net, device=None, find_unused_parameters=False, force=False):
    """DDP wrap like base_model.py:111-114.  find_unused_parameters defaults to False: every one of
    the 591 parameter tensors receives a gradient, so the graph walk is wasted work.
    `force`: wrap at world size 1 as well (one-GPU self-tests of the RCCL path: the reducer then runs its bucketed
    all-reduce over a one-rank communicator)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return net
    ids = [device.index] if device is not None and device.type == "cuda" else None
    # gradient_as_bucket_view: the 591 gradients are views of the reducer's flat bucket instead of being copied into it
    # one small kernel at a time (measured over a one-rank RCCL communicator, config 3: see DESIGN.md section 6)
    return torch.nn.parallel.DistributedDataParallel(net, device_ids=ids, gradient_as_bucket_view=True,
                                                     find_unused_parameters=find_unused_parameters)


def train_step(net, optimizer, lq, gt, as_float=True):
    """One optimize_parameters(): zero_grad, forward, L1 + 0.1*FFT, backward (DDP all-reduce), step.
    Returns the reduced losses {name: python float} like the reference's `reduce_loss_dict` (base_model.py:376-401: a log dict
    of floats - callers format / json-dump it).  as_float=False returns 0-dim DEVICE tensors instead: nothing in the step then
    waits for the GPU (a `float()` is a host synchronisation per iteration, under DDP on every rank); `loss_values()` converts
    such a result when it is time to log (the reference logs every `print_freq` iterations).  bench.py and the tools time the
    step with as_float=False."""
    optimizer.zero_grad(set_to_none=True)
    out = net(lq)
    l_pix, l_freq = losses(out, gt)
    (l_pix + l_freq).mean().backward()
    optimizer.step()
    return reduce_loss_dict({"l_pix": l_pix.detach(), "l_freq": l_freq.detach()}, as_float=as_float)


class GraphedTrainStep:
    """train_step() captured ONCE into a HIP graph (forward, both losses, backward, AdamW) and replayed: a BASELINE config-3 step is
    ~2,500 kernel launches behind ~50 ms of Python and autograd bookkeeping - as much as the GPU needs for the kernels (54 ms), so on a
    host with slower cores the eager step waits for the host (60.6 against 54.8 ms of kernels on one box of the pool, round 5); a
    replay costs the host one call.  Single-process training only (GraphedDDPTrainStep is the data-parallel form: a
    DistributedDataParallel reducer cannot run inside a capture); fixed batch shape; `optimizer` must be make_optimizer(..., capturable=True): its learning rate is a device
    tensor, so a torch LR scheduler stepped between replays takes effect (it fills the tensor in place; a float lr is refused -
    it would be frozen into the graph).  The warm-up steps are real optimizer steps at the optimizer's current lr.

        step = GraphedTrainStep(net, optimizer, lq0, gt0)          # 3 eager warm-up steps on (lq0, gt0), then the capture
        losses = step(lq, gt)                                      # copies the batch into the graph's input buffers, replays
    Returns {"l_pix", "l_freq"} as 0-dim device tensors of the step just replayed (loss_values() to log them)."""

    def __init__(self, net, optimizer, lq, gt, warmup=3):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            raise RuntimeError("GraphedTrainStep: single-process training only (under torch.distributed: GraphedDDPTrainStep)")
        if not all(g.get("capturable", False) for g in optimizer.param_groups):
            raise RuntimeError("GraphedTrainStep: the optimizer must be capturable (make_optimizer(net, capturable=True))")
        if not all(isinstance(g["lr"], torch.Tensor) and g["lr"].is_cuda for g in optimizer.param_groups):
            raise RuntimeError("GraphedTrainStep: the optimizer's lr must be a device tensor (make_optimizer(net, capturable=True)): "
                               "a float lr is frozen into the captured graph and learning-rate schedules would be ignored")
        self.lq, self.gt = lq.clone(), gt.clone()
        side = torch.cuda.Stream(lq.device)
        side.wait_stream(torch.cuda.current_stream(lq.device))
        with torch.cuda.stream(side):                              # warm-up off the capture's stream (allocator pools, library set-up)
            for _ in range(warmup):
                train_step(net, optimizer, self.lq, self.gt, as_float=False)
        torch.cuda.current_stream(lq.device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        mode = {}
        if dist.is_available() and dist.is_initialized():
            # a (one-rank) process group exists: its watchdog thread polls events while this thread captures, which the default
            # global capture mode turns into a fatal error in the watchdog (see GraphedDDPTrainStep)
            torch.cuda.synchronize(lq.device)
            mode = {"capture_error_mode": "thread_local"}
        with torch.cuda.graph(self.graph, **mode):
            out = net(self.lq)
            l_pix, l_freq = losses(out, self.gt)
            (l_pix + l_freq).mean().backward()
            optimizer.step()
        self.losses = {"l_pix": l_pix.detach(), "l_freq": l_freq.detach()}

    def __call__(self, lq=None, gt=None):
        if lq is not None:
            self.lq.copy_(lq)
        if gt is not None:
            self.gt.copy_(gt)
        self.graph.replay()
        return self.losses


class GraphedDDPTrainStep:
    """The data-parallel optimize_parameters() (femasr_model.py:157-185 under the DistributedDataParallel wrap of
    base_model.py:111-114) with the host taken out of the step: every rank replays

        graph A   forward, both losses, backward, the 591 gradients (pre-divided by the world size, as DDP's reducer does)
                  and the two losses copied into ONE flat fp32 buffer
        exchange  ONE all-reduce (sum) of that buffer - RCCL over xGMI: 6.05 MB per step for the shipped config
        graph B   AdamW on gradients that ARE views of the buffer

    instead of ~2,400 eager launches behind ~50 ms of Python, autograd and reducer bookkeeping per step and rank - with eight
    ranks on one host that work runs on whatever cores the host has left per rank (DESIGN.md section 6 has the measured host
    cost per step).  `net` is the BARE module (not the DDP wrap: the reducer's autograd hooks cannot run inside a capture); the
    constructor broadcasts rank 0's parameters and buffers like DDP's does.  `collective`:
        "split"     the all-reduce is an ordinary call between the two replays (any backend, gloo included)
        "captured"  one graph holds A, the all-reduce and B (backend nccl = RCCL only: its collectives are stream-ordered
                    kernels and capture like any other launch)
    `capture=False` runs the same three phases eagerly - what the CPU / gloo tests exercise, and the cross-check of the replays.
    Fixed batch shape; `optimizer` as for GraphedTrainStep on a GPU.  Returns {"l_pix", "l_freq"}: 0-dim tensors holding the
    MEAN over ranks of the step just run (every rank has them - they ride in the gradient buffer; the reference's
    reduce_loss_dict, base_model.py:376-401, leaves them on rank 0 only)."""

    def __init__(self, net, optimizer, lq, gt, warmup=3, process_group=None, capture=True, collective="split"):
        if isinstance(net, (torch.nn.parallel.DistributedDataParallel, torch.nn.DataParallel)):
            raise RuntimeError("GraphedDDPTrainStep: pass the bare module, not its DistributedDataParallel wrap")
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("GraphedDDPTrainStep: torch.distributed is not initialised (single process: GraphedTrainStep)")
        if collective not in ("split", "captured"):
            raise ValueError("collective: 'split' or 'captured'")
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.net, self.optimizer, self.capture, self.collective = net, optimizer, capture, collective
        self.params = [p for g in optimizer.param_groups for p in g["params"]]
        if capture:
            if not all(p.is_cuda for p in self.params):
                raise RuntimeError("GraphedDDPTrainStep: capture needs the network on a GPU (capture=False runs eagerly)")
            if not all(g.get("capturable", False) and isinstance(g["lr"], torch.Tensor) and g["lr"].is_cuda
                       for g in optimizer.param_groups):
                raise RuntimeError("GraphedDDPTrainStep: the optimizer must be make_optimizer(net, capturable=True)")
            if collective == "captured" and dist.get_backend(process_group) != "nccl":
                raise RuntimeError("GraphedDDPTrainStep: collective='captured' needs the nccl (RCCL) backend")
        # DDP's constructor: every rank starts from rank 0's parameters and buffers
        with torch.no_grad():
            for t in list(net.parameters()) + list(net.buffers()):
                dist.broadcast(t, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0,
                               group=process_group)
        dev = self.params[0].device
        self.sizes = [p.numel() for p in self.params]
        self.flat = torch.zeros(sum(self.sizes) + 2, dtype=torch.float32, device=dev)     # gradients | l_pix | l_freq
        self.views = [v.view_as(p) for v, p in zip(self.flat[:-2].split(self.sizes), self.params)]
        self.losses = {"l_pix": self.flat[-2], "l_freq": self.flat[-1]}
        self.lq, self.gt = lq.clone(), gt.clone()
        if not capture:
            return
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                              # warm-up off the capture's stream: real steps, all ranks alike
            for _ in range(warmup):
                self._produce()
                self._exchange()
                self._update()
        torch.cuda.current_stream(dev).wait_stream(side)
        # no collective of the warm-up may still be in flight, and the captures are THREAD-LOCAL: ProcessGroupNCCL's watchdog
        # thread polls its work events (hipEventQuery) while this thread captures; under the default global capture mode that
        # call is an error in the watchdog and ends the process
        torch.cuda.synchronize(dev)
        mode = {"capture_error_mode": "thread_local"}
        if collective == "captured":
            self.graph_a, self.graph_b = torch.cuda.CUDAGraph(), None
            with torch.cuda.graph(self.graph_a, **mode):
                self._produce()
                self._exchange()
                self._update()
        else:
            self.graph_a, self.graph_b = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_a, **mode):
                self._produce()
            with torch.cuda.graph(self.graph_b, pool=self.graph_a.pool(), **mode):
                self._update()

    def _produce(self):
        for p in self.params:
            p.grad = None
        out = self.net(self.lq)
        l_pix, l_freq = losses(out, self.gt)
        (l_pix + l_freq).mean().backward()
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.params]
        torch._foreach_copy_(self.views, grads)
        self.flat[-2].copy_(l_pix.detach())
        self.flat[-1].copy_(l_freq.detach())
        self.flat.mul_(1.0 / self.world)                           # DDP's reducer divides before it sums

    def _exchange(self):
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)

    def _update(self):
        for p, v in zip(self.params, self.views):
            p.grad = v
        self.optimizer.step()

    def __call__(self, lq=None, gt=None):
        if lq is not None:
            self.lq.copy_(lq)
        if gt is not None:
            self.gt.copy_(gt)
        if not self.capture:
            self._produce()
            self._exchange()
            self._update()
        elif self.graph_b is None:
            self.graph_a.replay()
        else:
            self.graph_a.replay()
            self._exchange()
            self.graph_b.replay()
        return self.losses


def loss_values(loss_dict):
    """{name: python float} of a train_step() result (synchronises: call it when logging, not every step)."""
    return {k: float(v) for k, v in loss_dict.items()}


def reduce_loss_dict(loss_dict, as_float=True):
    """base_model.py:376-401: sum-reduce to rank 0 then divide by world size (the values mean something on rank 0 only,
    as in the reference).  as_float=False keeps 0-dim tensors on their device (no host synchronisation)."""
    conv = (lambda v: float(v)) if as_float else (lambda v: v)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return {k: conv(v) for k, v in loss_dict.items()}
    keys = sorted(loss_dict)
    vec = torch.stack([loss_dict[k].float() for k in keys])
    dist.reduce(vec, dst=0)
    if dist.get_rank() == 0:
        vec /= dist.get_world_size()
    return {k: conv(v) for k, v in zip(keys, vec.unbind(0))}


# ---- reference-format checkpoints (basicsr/models/base_model.py:214-261, :299-326, :328-373) -----------------------
def _bare(net):
    return net.module if isinstance(net, (torch.nn.parallel.DistributedDataParallel, torch.nn.DataParallel)) else net


def _atomic_save(obj, path):
    """torch.save into a sibling temp file, then rename: a reader never sees a torn checkpoint (the reference retries
    the write three times instead, :247-260)."""
    import os
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    tmp = f"{path}.tmp{os.getpid()}"
    torch.save(obj, tmp)
    os.replace(tmp, path)


def save_network(net, path, current_iter=-1, epoch=0, param_key="params"):
    """`{param_key: state_dict on the CPU without 'module.' prefixes, 'iter': int | 'latest', 'epoch': int}` -
    the file layout `inference_wavemamba.py:74` / `load_network` read.  Rank 0 only (`@master_only`, :214)."""
    if dist.is_available() and dist.is_initialized() and dist.get_rank() != 0:
        return None
    state = {(k[7:] if k.startswith("module.") else k): v.detach().cpu() for k, v in _bare(net).state_dict().items()}
    _atomic_save({param_key: state, "iter": "latest" if current_iter == -1 else current_iter, "epoch": epoch}, path)
    return path


def load_network(net, path, strict=True, param_key="params", weights_only=True):
    """Counterpart of base_model.py:299-326: `param_key` falls back to 'params' when absent (None = the file is the
    bare state dict), 'module.' prefixes are dropped, and with strict=False tensors whose shape differs are skipped
    instead of raising.  Returns (missing_keys, unexpected_keys, skipped_for_shape).
    Reference-format files hold only tensors, ints, strs and (Ordered)dicts, so they load with `weights_only=True`
    (no arbitrary pickle code from a third-party checkpoint); pass weights_only=False for a trusted legacy file that
    pickles other objects."""
    blob = torch.load(path, map_location="cpu", weights_only=weights_only)
    if param_key is not None:
        if param_key not in blob and "params" in blob:
            param_key = "params"
        blob = blob[param_key]
    state = {(k[7:] if k.startswith("module.") else k): v for k, v in blob.items()}
    target = _bare(net)
    skipped = []
    if not strict:
        own = target.state_dict()
        skipped = sorted(k for k, v in state.items() if k in own and own[k].shape != v.shape)
        for k in skipped:
            del state[k]
    res = target.load_state_dict(state, strict=strict)
    from . import ops
    ops.conv2d_cache_clear()                 # prepared weight copies of the inference convolutions
    return list(res.missing_keys), list(res.unexpected_keys), skipped


def save_training_state(path, epoch, current_iter, optimizers, schedulers=()):
    """`{'epoch', 'iter', 'optimizers': [...], 'schedulers': [...]}` (base_model.py:328-357); nothing is written for
    current_iter == -1, and only rank 0 writes."""
    if current_iter == -1 or (dist.is_available() and dist.is_initialized() and dist.get_rank() != 0):
        return None
    _atomic_save({"epoch": epoch, "iter": current_iter, "optimizers": [o.state_dict() for o in optimizers],
                  "schedulers": [s.state_dict() for s in schedulers]}, path)
    return path


def resume_training(path_or_state, optimizers, schedulers=(), weights_only=True):
    """base_model.py:359-373.  Returns (epoch, iter).  `weights_only` as in load_network."""
    st = path_or_state if isinstance(path_or_state, dict) else torch.load(path_or_state, map_location="cpu",
                                                                            weights_only=weights_only)
    if len(st["optimizers"]) != len(optimizers) or len(st["schedulers"]) != len(schedulers):
        raise ValueError("resume_training: optimizer / scheduler count differs from the saved state")
    for o, s in zip(optimizers, st["optimizers"]):
        o.load_state_dict(s)
    for o, s in zip(schedulers, st["schedulers"]):
        o.load_state_dict(s)
    return st["epoch"], st["iter"]
