#!/usr/bin/env python3
"""BASELINE config 3: Wave-Mamba UHD-LL training step, batch 8 synthetic 512x512 pairs per GPU,
DistributedDataParallel over the GPUs of one node (RCCL gradient all-reduce only).

    python tools/train_bench.py [--steps 5 --warmup 2 --batch 8 --size 512]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P tools/train_bench.py --gpus N

One step = reference optimize_parameters (femasr_model.py:157-185): zero_grad, forward, L1 + 0.1 FFT-L1,
backward (selective-scan / DWT / IWT backward in HIP; DDP all-reduce of the 6.05 MB fp32 gradients),
AdamW step.  Prints one JSON line on rank 0: training images/s over all ranks (weak scaling)."""
import argparse, json, os, sys, time
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wave_mamba_amd as wm
import bench                       # rank plumbing shared with bench.py (tests/test_bench_ranks.py drives it under gloo)

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, default=1)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--size", type=int, default=512)
args = ap.parse_args()
rank, world, local = bench.rank_env()
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
if world > 1:
    dist.init_process_group("nccl", device_id=dev)
torch.manual_seed(0)
net = wm.WaveMamba(in_chn=3, wf=32, n_l_blocks=[1, 2, 4], n_h_blocks=[1, 1, 2], ffn_scale=2.0).train().to(dev)
ddp = wm.trainer.wrap_ddp(net, dev)
opt = wm.trainer.make_optimizer(ddp)
g = torch.Generator().manual_seed(bench.image_seed(rank))
lq = torch.rand(args.batch, 3, args.size, args.size, generator=g).to(dev)
gt = torch.rand(args.batch, 3, args.size, args.size, generator=g).to(dev)
state = {}


def step():
    state["losses"] = wm.trainer.train_step(ddp, opt, lq, gt, as_float=False)


for _ in range(args.warmup):
    step()
wm.ops.prof_enable(True)
el = bench.timed_steps(step, args.steps, 0, torch.cuda.synchronize, dist.barrier if world > 1 else (lambda: None))
prof = wm.ops.prof_collect(); wm.ops.prof_enable(False)
el = bench.max_over_ranks(el, world, dev)
losses = wm.trainer.loss_values(state["losses"])
if rank == 0:
    print(json.dumps({"metric": "training images/sec (UHD-LL config, 512x512 crops)", "value": bench.whole_job_value(world, args.steps, args.batch, el),
                      "unit": "images/s", "n_gpus": world, "steps": args.steps, "ms_per_step": 1e3 * el / args.steps,
                      "scaling": "weak", "dtype": "f32", "data": "synthetic",
                      "config": {"workload": f"batch {args.batch} x 3x{args.size}x{args.size} per GPU, L1 + 0.1 FFT loss, AdamW, DDP"},
                      "losses": losses, "peak_mem_GB": torch.cuda.max_memory_allocated() / 2**30,
                      "kernel_ms_per_step": {k: v[1] / args.steps for k, v in prof.items() if v[0]}}))
if world > 1:
    dist.destroy_process_group()
