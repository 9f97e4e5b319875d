"""CPU, gloo, world sizes 2 and 4: the rank plumbing of bench.py and tools/train_bench.py - env parsing, the barrier /
sync bracket around exactly K timed steps, the max-over-ranks reduction, the whole-job value, per-rank inputs and the
rank-0-only legs - driven with a fake step (the real step needs a GPU; no 1 -> 8 GPU curve has been measured yet, this
covers the control flow the driver's multi-GPU launch will take)."""
import os
import socket
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import bench
    r, w, lr = bench.rank_env()
    assert (r, w, lr) == (rank, world, rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        calls, syncs, barriers = [], [], []

        def step():
            time.sleep(0.004 * (rank + 1))          # rank r is (r + 1) x slower: the job runs at the slowest rank's pace
            calls.append(time.perf_counter())

        def barrier():
            barriers.append(len(calls))
            dist.barrier()
        el = bench.timed_steps(step, steps=6, warmup=2, sync=lambda: syncs.append(len(calls)), barrier=barrier)
        assert len(calls) == 8                       # W warm-up + exactly K timed
        assert syncs == [2, 8] and barriers == [2, 8]        # bracket: after the warm-up, after the K-th step
        el_max = bench.max_over_ranks(el, world, "cpu")
        value = bench.whole_job_value(world, 6, 1, el_max)
        legs_rank0_only = (rank == 0 and world == 1)         # cpu_baseline / parity / concurrent legs: N = 1 only
        torch.save({"el": el, "el_max": el_max, "value": value, "seed": bench.image_seed(rank), "legs": legs_rank0_only},
                   os.path.join(out_dir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_bench_rank_logic_gloo(world, tmp_path):
    mp.spawn(_worker, args=(world, free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(tmp_path / f"r{r}.pt") for r in range(world)]
    mx = res[0]["el_max"]
    assert all(abs(r["el_max"] - mx) < 1e-12 for r in res), "every rank must hold the same max-over-ranks time"
    # the barriers on both sides make every rank's own bracket span the slowest rank's steps
    assert mx >= 6 * 0.004 * world * 0.95
    assert abs(mx - max(r["el"] for r in res)) < 1e-12
    assert all(abs(r["value"] - world * 6 / mx) < 1e-9 for r in res)          # whole-job aggregate, not per GPU
    assert sorted(r["seed"] for r in res) == [1234 + r for r in range(world)]  # every replica its own image
    assert not any(r["legs"] for r in res)


def test_single_process_needs_no_process_group():
    import bench
    assert bench.rank_env({}) == (0, 1, 0)
    assert bench.max_over_ranks(1.25, 1, "cpu") == 1.25
    n = []
    el = bench.timed_steps(lambda: n.append(1), 3, 1, lambda: None, lambda: None)
    assert len(n) == 4 and el >= 0
    assert bench.whole_job_value(1, 10, 1, 0.4) == 25.0
    assert bench.scan_positions(2176, 3840) == 7311360           # positions scanned per UHD image


def test_train_bench_shares_the_rank_helpers():
    """tools/train_bench.py times its steps through the same bracket and reduction."""
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "train_bench.py")).read()
    for name in ("bench.rank_env", "bench.timed_steps", "bench.max_over_ranks", "bench.whole_job_value"):
        assert name in src, name
