#!/usr/bin/env python3
"""What a rank's step costs the HOST, and what happens to the step when the rank has few cores.

Eight ranks on one host share its cores (the round-5 bench box offered 16 usable cores: two per rank).  For each of
  inference   one UHD forward (1 x 3 x 2176 x 3840, shipped config, eager launches through ops / ctypes)
  train       one optimize_parameters() of BASELINE config 3 (8 x 3 x 512 x 512), eager
  train-graph the same step replayed from a HIP graph (trainer.GraphedTrainStep)
this prints the wall time per step and the CPU time the process burnt per step (all its threads: time.process_time), with the
process pinned to all / 2 / 1 cores (os.sched_setaffinity before torch is imported, one subprocess per row).  A step is
host-bound on k cores when its wall time there exceeds the all-cores wall time; CPU ms per step is the lower bound of the wall
time on one core.

    python tools/host_bound.py            (GPU box)  ->  table on stdout
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(leg, ncores, steps):
    avail = sorted(os.sched_getaffinity(0))
    if ncores:
        os.sched_setaffinity(0, set(avail[:ncores]))
    sys.path.insert(0, ROOT)
    import torch
    import bench
    import wave_mamba_amd as wm
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if leg == "inference":
        net = bench.build_model(dev)
        x = bench.pad_to(torch.rand(1, 3, 2160, 3840, generator=torch.Generator().manual_seed(1))).to(dev)

        def step():
            with torch.no_grad():
                net.restoration_network(x)
        units = 1
    else:
        torch.manual_seed(0)
        net = wm.WaveMamba(**bench.SHIPPED).train().to(dev)
        g = torch.Generator().manual_seed(2)
        lq, gt = torch.rand(8, 3, 512, 512, generator=g).to(dev), torch.rand(8, 3, 512, 512, generator=g).to(dev)
        if leg == "train":
            opt = wm.trainer.make_optimizer(net)

            def step():
                wm.trainer.train_step(net, opt, lq, gt, as_float=False)
        else:
            gs = wm.trainer.GraphedTrainStep(net, wm.trainer.make_optimizer(net, capturable=True), lq, gt)

            def step():
                gs()
        units = 8
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    c0, t0 = time.process_time(), time.perf_counter()
    for _ in range(steps):
        step()
    t_issue = time.perf_counter() - t0                      # the host is done issuing; the GPU may still be running
    torch.cuda.synchronize()
    t1, c1 = time.perf_counter(), time.process_time()
    print(json.dumps({"leg": leg, "cores": len(os.sched_getaffinity(0)), "wall_ms": 1e3 * (t1 - t0) / steps,
                      "issue_ms": 1e3 * t_issue / steps, "cpu_ms": 1e3 * (c1 - c0) / steps,
                      "units_per_s": units * steps / (t1 - t0)}))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
    steps = int(os.environ.get("WM_HOST_BOUND_STEPS", "10"))
    print(f"usable cores on this box: {len(os.sched_getaffinity(0))} of {os.cpu_count()}")
    print(f"{'leg':12s} {'cores':>5s} {'wall ms/step':>13s} {'host issue ms/step':>19s} {'CPU ms/step':>12s} {'units/s':>9s}")
    rows = []
    for leg in ("inference", "train", "train-graph"):
        for nc in (0, 2, 1):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", leg, str(nc), str(steps)],
                               capture_output=True, text=True, timeout=900)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(f"{leg:12s} {nc:5d}  failed: {r.stderr[-300:]}")
                continue
            d = json.loads(line[-1])
            rows.append(d)
            print(f"{d['leg']:12s} {d['cores']:5d} {d['wall_ms']:13.2f} {d['issue_ms']:19.2f} {d['cpu_ms']:12.2f} {d['units_per_s']:9.2f}")
    print(json.dumps(rows))


if __name__ == "__main__":
    main()
