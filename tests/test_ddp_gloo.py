"""CPU, world sizes 2 and 4 over gloo: the N > 1 paths.

  * training: DistributedDataParallel over image batches, one gradient all-reduce per step; the
    averaged gradients must equal the single-process gradients on the concatenated batch
    ("distributed" row).  Hot-path ops: the CPU oracle installed as test backend.
  * inference: replicas - each rank an independent image, no data-path collective; only the
    max-over-ranks timing reduction bench.py performs.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

CFG = dict(in_chn=3, wf=8, n_l_blocks=[1, 1, 1], n_h_blocks=[1, 1, 1], ffn_scale=2.0)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    import wave_mamba_amd as wm
    from wave_mamba_amd.archs import wavemamba_arch as arch
    from oracle import oracle
    from oracle import backend as oracle_backend
    oracle.set_num_threads(2)
    oracle_backend.set_ops_backend(oracle)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        net = wm.WaveMamba(**CFG).train()
        ddp = wm.trainer.wrap_ddp(net)
        assert isinstance(ddp, torch.nn.parallel.DistributedDataParallel)
        opt = wm.trainer.make_optimizer(ddp)
        g = torch.Generator().manual_seed(1234)
        lq, gt = torch.rand(4, 3, 32, 32, generator=g), torch.rand(4, 3, 32, 32, generator=g)
        per = 4 // world
        shard = slice(rank * per, rank * per + per)               # EnlargedSampler-style rank shard
        opt.zero_grad()
        out = ddp(lq[shard])
        l_pix, l_freq = wm.trainer.losses(out, gt[shard])
        (l_pix + l_freq).backward()
        grads = {k: p.grad.clone() for k, p in net.named_parameters()}
        red = wm.trainer.reduce_loss_dict({"l_pix": l_pix.detach(), "l_freq": l_freq.detach()})
        opt.step()
        # replicas: independent forward per rank, then the max-over-ranks timing reduction of bench.py
        with torch.no_grad():
            y = net(torch.rand(1, 3, 32, 32, generator=torch.Generator().manual_seed(100 + rank)))
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        torch.save({"grads": grads, "loss": red, "tmax": float(t), "ysum": float(y.sum()),
                    "w0": net.restoration_network.last.weight.detach().clone()},
                   os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 4])
def test_ddp_ranks_match_single_process(world, tmp_path):
    mp.spawn(_worker, args=(world, free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(tmp_path / f"rank{r}.pt") for r in range(world)]
    r0 = res[0]
    # gradients are identical on every rank after the all-reduce, and so are the updated weights
    for r in res[1:]:
        for k in r0["grads"]:
            assert torch.equal(r0["grads"][k], r["grads"][k]), k
        assert torch.equal(r0["w0"], r["w0"])
        assert r["tmax"] == r0["tmax"] == float(world)
    assert len({r["ysum"] for r in res}) == world                # replicas saw different images

    # single process on the concatenated batch: mean-reduced losses => same gradients
    import wave_mamba_amd as wm
    from wave_mamba_amd.archs import wavemamba_arch as arch
    from oracle import oracle
    from oracle import backend as oracle_backend
    prev = oracle_backend.set_ops_backend(oracle)
    try:
        torch.manual_seed(0)
        net = wm.WaveMamba(**CFG).train()
        g = torch.Generator().manual_seed(1234)
        lq, gt = torch.rand(4, 3, 32, 32, generator=g), torch.rand(4, 3, 32, 32, generator=g)
        l_pix, l_freq = wm.trainer.losses(net(lq), gt)
        (l_pix + l_freq).backward()
    finally:
        oracle_backend.set_ops_backend(prev)
    worst = 0.0
    for k, p in net.named_parameters():
        a, b = r0["grads"][k].double(), p.grad.double()
        worst = max(worst, float((a - b).norm() / b.norm().clamp_min(1e-12)))
    assert worst < 1e-4, f"DDP vs single-process gradient mismatch {worst:.3e}"
    assert abs(r0["loss"]["l_pix"] - float(l_pix.detach())) < 1e-5


def _flat_step_worker(rank, world, port, out_dir):
    """Two optimize_parameters() per rank: (a) DistributedDataParallel + trainer.train_step, (b) trainer.GraphedDDPTrainStep in
    its eager form (capture=False: the same three phases the GPU replays - gradients and losses into one flat buffer, ONE
    all-reduce, AdamW on views of the buffer)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    import wave_mamba_amd as wm
    from oracle import oracle
    from oracle import backend as oracle_backend
    oracle.set_num_threads(2)
    oracle_backend.set_ops_backend(oracle)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(77)
        lq, gt = torch.rand(2, 4, 3, 32, 32, generator=g), torch.rand(2, 4, 3, 32, 32, generator=g)    # [step][image]
        per = 4 // world
        shard = slice(rank * per, rank * per + per)
        torch.manual_seed(0)
        net_a = wm.WaveMamba(**CFG).train()
        torch.manual_seed(100 + rank)              # (b) starts from rank-dependent weights: the constructor must broadcast rank 0's
        net_b = wm.WaveMamba(**CFG).train()
        ddp = wm.trainer.wrap_ddp(net_a)
        opt_a = wm.trainer.make_optimizer(ddp)
        if rank == 0:
            net_b.load_state_dict(net_a.state_dict())
        opt_b = wm.trainer.make_optimizer(net_b)
        step_b = wm.trainer.GraphedDDPTrainStep(net_b, opt_b, lq[0, shard], gt[0, shard], capture=False)
        with pytest.raises(RuntimeError):
            wm.trainer.GraphedDDPTrainStep(ddp, opt_a, lq[0, shard], gt[0, shard], capture=False)
        rec = []
        for s in range(2):
            la = wm.trainer.train_step(ddp, opt_a, lq[s, shard], gt[s, shard], as_float=False)
            lb = step_b(lq[s, shard], gt[s, shard])
            rec.append({"loss_a": {k: float(v) for k, v in la.items()}, "loss_b": {k: float(v) for k, v in lb.items()},
                        "grad_a": {k: p.grad.clone() for k, p in net_a.named_parameters()},
                        "grad_b": {k: p.grad.clone() for k, p in net_b.named_parameters()},
                        "w_a": {k: p.detach().clone() for k, p in net_a.named_parameters()},
                        "w_b": {k: p.detach().clone() for k, p in net_b.named_parameters()}})
        torch.save(rec, os.path.join(out_dir, f"flat{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 4])
def test_flat_bucket_step_equals_ddp_step(world, tmp_path):
    """the graph-replayed data-parallel step (trainer.GraphedDDPTrainStep) against the eager
    DistributedDataParallel step, over gloo.  World 2: BIT-equal gradients and parameters after each of two steps (g0/2 + g1/2 is
    one rounding whatever the reduction's order); world 4: to 1e-6 of each tensor's largest entry (gloo's ring sums a bucket's
    chunks in an order that depends on where the element sits in its bucket, and the two paths bucket differently)."""
    mp.spawn(_flat_step_worker, args=(world, free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(tmp_path / f"flat{r}.pt") for r in range(world)]
    for s in range(2):
        r0 = res[0][s]
        for r in res:
            for k in r0["w_a"]:
                assert torch.equal(r[s]["w_b"][k], r0["w_b"][k]), (s, k)                 # ranks stay in lock step
                if world == 2:
                    assert torch.equal(r[s]["grad_a"][k], r[s]["grad_b"][k]), (s, k)
                    assert torch.equal(r[s]["w_a"][k], r[s]["w_b"][k]), (s, k)
                else:
                    ga, gb = r[s]["grad_a"][k], r[s]["grad_b"][k]
                    assert float((ga - gb).abs().max()) <= 1e-6 * float(ga.abs().max()) + 1e-12, (s, k)
            # every rank holds the mean loss; the reference's reduce leaves it on rank 0 only
            for k in ("l_pix", "l_freq"):
                assert abs(r[s]["loss_b"][k] - r0["loss_a"][k]) <= 1e-6 * abs(r0["loss_a"][k]), (s, k)
