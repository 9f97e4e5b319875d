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
        # the identity block of the N > 1 line: every rank's device gathered over the job's own process group
        ident = {"index": rank, "uuid": None, "pci": f"0000:{rank:02x}:00", "pid": os.getpid()}
        idents = bench.gather_identities(ident, world)
        assert [i["index"] for i in idents] == list(range(world)) and bench.distinct_devices(idents)
        assert not bench.distinct_devices(idents + [dict(idents[0])])          # two ranks on one GPU are detected
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


def test_bench_self_launch_starts_one_rank_per_gpu(tmp_path):
    """`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset) re-executes itself under
    torch.distributed.run with N ranks on 127.0.0.1 (round 3's bench died on `assert world == args.gpus` there).  The
    launcher is exercised for real on CPU with a stand-in script that does what bench.main does first: read the rank
    environment, join the process group, reduce a per-rank value."""
    import subprocess
    import sys
    import bench
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "3"], port=1234)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "1234"
    assert cmd[-5] == os.path.abspath(bench.__file__) and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rank_probe.py"
    script.write_text(
        "import os, sys, torch, torch.distributed as dist\n"
        f"sys.path.insert(0, {root!r})\n"
        "import bench\n"
        "rank, world, local = bench.rank_env()\n"
        "assert world == int(sys.argv[sys.argv.index('--gpus') + 1])\n"
        "dist.init_process_group('gloo')\n"
        "t = bench.max_over_ranks(1.0 + rank, world, 'cpu')\n"
        f"open(os.path.join({str(tmp_path)!r}, f'rank{{rank}}.txt'), 'w').write(f'{{world}} {{local}} {{t}}')\n"
        "dist.destroy_process_group()\n")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    rc = subprocess.call(bench.launch_command(2, ["--gpus", "2"], script=str(script)), env=env, timeout=300)
    assert rc == 0
    assert [open(tmp_path / f"rank{r}.txt").read() for r in range(2)] == ["2 0 2.0", "2 1 2.0"]
    # main() takes that branch before it touches the GPU
    src = open(os.path.join(root, "bench.py")).read()
    assert src.index('"WORLD_SIZE" not in os.environ') < src.index("torch.cuda.is_available()")
