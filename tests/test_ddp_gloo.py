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
