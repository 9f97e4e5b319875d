#!/usr/bin/env python3
"""bench.py - UHD (3840x2160) images/s of the Wave-Mamba forward on MI355X, with the roofline
fraction of the dominant hot-path kernel and the CPU-oracle baseline timed beside it.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one forward of the shipped WaveMamba (inference_wavemamba.py:71-75 config, seeded random
init - checkpoints are not distributed) over one synthetic 1x3x2160x3840 image, reflect-padded to
2176x3840 exactly as the reference's inference script does (:28-36), input already resident in HBM.
N > 1: one process per GPU, each an independent replica on its own image (the path shards by image,
no data-path collective) -> weak scaling; value = N*K images / max-over-ranks time.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the field definitions).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import wave_mamba_amd as wm                                     # noqa: E402
from wave_mamba_amd.archs import wavemamba_arch as arch        # noqa: E402

SHIPPED = dict(in_chn=3, wf=32, n_l_blocks=[1, 2, 4], n_h_blocks=[1, 1, 2], ffn_scale=2.0)
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
SCAN_BYTES_PER_POS = {16: 3584, 32: 4096}   # 4*(3*KD + 2*K*N), KD = 256, K = 4
# kernel classes whose HIP events are recorded inside the timed region (the candidates for `roofline`)
TIMED_PROF = ("ss2d_row_scan", "ss2d_col_scan")


def pad_to(x, mult=128):
    """inference_wavemamba.py:28-36: reflect-pad bottom/right to a multiple of 128."""
    h, w = x.shape[-2:]
    return F.pad(x, (0, (mult - w % mult) % mult, 0, (mult - h % mult) % mult), "reflect")


def scan_positions(h, w):
    """positions scanned per image: 2*(L1 + 2*L2 + 4*L3) for n_l_blocks [1,2,4]."""
    l1, l2, l3 = (h // 2) * (w // 2), (h // 4) * (w // 4), (h // 8) * (w // 8)
    return 2 * (l1 + 2 * l2 + 4 * l3)


def build_model(device):
    torch.manual_seed(0)
    return wm.WaveMamba(**SHIPPED).eval().to(device)


def cpu_baseline(sample_hw, uhd_hw, repeats):
    """The CPU oracle ("port") inside the same network on host cores, on a bounded sample."""
    from oracle import oracle
    from oracle import backend as oracle_backend
    cores = oracle.usable_cpus()
    torch.set_num_threads(cores)
    oracle.set_num_threads(cores)
    net = build_model("cpu")
    prev = oracle_backend.set_ops_backend(oracle)
    try:
        with torch.no_grad():
            # bounded: shrink the sample until one forward takes < ~8 s on this host
            while True:
                x = torch.rand(1, 3, *sample_hw, generator=torch.Generator().manual_seed(1234))
                t0 = time.perf_counter()
                y = net.restoration_network(x)                  # warm-up / probe
                probe = time.perf_counter() - t0
                if probe < 8.0 or min(sample_hw) <= 128:
                    break
                sample_hw = (sample_hw[0] // 2, sample_hw[1] // 2)
            times = [probe] if probe > 8.0 else []
            for _ in range(0 if times else repeats):
                t0 = time.perf_counter()
                y = net.restoration_network(x)
                times.append(time.perf_counter() - t0)
    finally:
        oracle_backend.set_ops_backend(prev)
    repeats = len(times)
    t = sorted(times)[len(times) // 2]
    scale = (uhd_hw[0] * uhd_hw[1]) / (sample_hw[0] * sample_hw[1])
    return {
        "value": 1.0 / (t * scale), "unit": "images/s", "cores": cores, "host_cpus": os.cpu_count(), "kind": "port",
        "sample": f"1x3x{sample_hw[0]}x{sample_hw[1]} fp32 forward (median of {repeats}, {t:.2f} s), same "
                  f"network with the C/OpenMP oracle as hot-path backend + PyTorch-CPU for the rest; "
                  f"scaled by the padded-area ratio {scale:.2f} to one 2176x3840 image",
    }, x, y


def scan_op_boundary(device, hp, wp, iters=3):
    """The drop-in selective_scan_fn (reference call signature, 3584 B/position) on the UHD level-1
    shape (B=1, KD=256, L=hp*wp/4): HBM fraction of its chunk-scan kernel and of the whole op."""
    L, dim, N, G = (hp // 2) * (wp // 2), 256, 16, 4
    g = torch.Generator(device=device).manual_seed(7)
    u = torch.randn(1, dim, L, device=device, generator=g)
    dl = 0.5 * torch.randn(1, dim, L, device=device, generator=g)
    A = -torch.arange(1, N + 1, device=device, dtype=torch.float32).repeat(dim, 1) * \
        torch.exp(0.2 * torch.randn(dim, N, device=device, generator=g))
    Bm, Cm = (torch.randn(1, G, N, L, device=device, generator=g) for _ in range(2))
    D = torch.randn(dim, device=device, generator=g)
    bias = 0.5 * torch.randn(dim, device=device, generator=g) - 4.0
    wm.ops.selective_scan_fn(u, dl, A, Bm, Cm, D, None, bias, True)
    torch.cuda.synchronize()
    wm.ops.prof_enable(True)
    for _ in range(iters):
        wm.ops.selective_scan_fn(u, dl, A, Bm, Cm, D, None, bias, True)
    prof = wm.ops.prof_collect()
    wm.ops.prof_enable(False)
    r, c, s = (prof[k][1] / iters for k in ("selscan_chunk_reduce", "selscan_carry", "selscan_chunk_scan"))
    nbytes = SCAN_BYTES_PER_POS[N] * L
    return {"shape": f"u,delta (1,{dim},{L}); B,C (1,{G},{N},{L})", "algorithmic_bytes": nbytes,
            "chunk_scan_ms": s, "chunk_scan_frac": nbytes / (s * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "reduce_ms": r, "carry_ms": c, "whole_op_frac": nbytes / ((r + c + s) * 1e-3) / 1e9 / HBM_PEAK_GBS}


def graph_replay(step, steps, device):
    """The same step captured once into a HIP graph (torch.cuda.CUDAGraph = hipGraph on ROCm) and
    replayed: removes the ~1000 host-side launches per forward (launch-bound stretches at the small
    pyramid levels).  Reported next to `value`, which stays the eager, event-instrumented run."""
    try:
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()                                              # warm the private-pool allocations
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = step()
        g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            g.replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return {"images_per_s": steps / dt, "ms_per_step": 1e3 * dt / steps, "checksum": float(out.double().sum())}
    except Exception as e:                                      # capture is best-effort, never fatal
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def bf16_autocast(net, x, steps):
    """BASELINE config 2 names bf16 inference; the reference itself is fp32-only (SURVEY: pure .bfloat16()
    fails, autocast works).  Here: torch.autocast(bfloat16) for everything OUTSIDE the hot path (MIOpen /
    hipBLASLt convs and GEMMs of the HFE branch); the HIP hot path keeps fp32 state and I/O.  Reported
    beside `value` (which stays the parity-exact fp32 run) with the PSNR against the fp32 output."""
    try:
        with torch.no_grad():
            ref = net.restoration_network(x)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                for _ in range(2):
                    out = net.restoration_network(x)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    out = net.restoration_network(x)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
        mse = float((out.float().clamp(0, 1) - ref.clamp(0, 1)).pow(2).mean())
        return {"images_per_s": steps / dt, "ms_per_step": 1e3 * dt / steps,
                "psnr_vs_fp32_db": float(10 * torch.log10(torch.tensor(1.0 / max(mse, 1e-20)))),
                "rel_l2_vs_fp32": float((out.float() - ref).norm() / ref.norm())}
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--streams", type=int, default=1,
                    help="forwards in flight in the timed region: step i runs on HIP stream i %% STREAMS (default 1: back "
                         "to back on one stream, so the per-kernel HIP-event durations behind `roofline` are undisturbed)")
    ap.add_argument("--concurrent", type=int, default=4,
                    help="extra untimed leg at N = 1: throughput with this many forwards in flight on separate HIP "
                         "streams, no event instrumentation (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="also time the step replayed from a HIP graph")
    ap.add_argument("--bf16", action="store_true", help="also time torch.autocast(bfloat16) outside the hot path")
    ap.add_argument("--cpu-sample", type=int, nargs=2, default=[512, 1024])
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path is HIP-only (no CPU fallback)")
    # self-test hook (one-GPU boxes): WM_BENCH_SHARE_GPU=1 puts every rank on cuda:0 over gloo, to exercise the
    # multi-rank control flow (barriers, max-over-ranks reduction, rank-0-only legs); never set by the driver
    share = os.environ.get("WM_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if share:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)          # RCCL on ROCm
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    # CPU baseline first (rank 0, N = 1 only), before the GPU pass
    cpu, parity = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, xs, ys = cpu_baseline(tuple(args.cpu_sample), (2176, 3840), repeats=3)

    net = build_model(device)
    img = torch.rand(1, 3, args.height, args.width, generator=torch.Generator().manual_seed(1234 + rank))
    x = pad_to(img.to(device))
    hp, wp = x.shape[-2:]

    if cpu is not None:   # parity of the HIP path vs the CPU-oracle path on the very same sample
        with torch.no_grad():
            yg = net.restoration_network(xs.to(device)).cpu()
        tgt = torch.rand(xs.shape, generator=torch.Generator().manual_seed(4321))

        def psnr(a, b):
            qa, qb = (a.clamp(0, 1) * 255).round(), (b.clamp(0, 1) * 255).round()
            return float(20 * torch.log10(255.0 / (qa - qb).pow(2).mean().sqrt()))
        parity = {"rel_l2_vs_cpu_oracle": float((yg - ys).norm() / ys.norm()),
                  "abs_dpsnr_db": abs(psnr(yg, tgt) - psnr(ys, tgt))}

    # Consecutive steps (independent images) may run on `--streams` HIP streams round-robin: up to that many forwards
    # are then in flight and the kernels of one fill the tails and the small launches of another (one stream: 23.0
    # images/s, four: 28.6 on one MI355X) - but every individual launch stretches, so the default timed region keeps
    # one stream and the concurrent rate is reported by the `concurrent_forwards` leg.
    nstreams = max(1, args.streams)
    streams = [torch.cuda.Stream(device) for _ in range(nstreams)] if nstreams > 1 else [torch.cuda.current_stream(device)]
    counter = [0]

    def step(single=False):
        st = streams[0] if single else streams[counter[0] % nstreams]
        counter[0] += 1
        with torch.no_grad(), torch.cuda.stream(st):
            out = net.restoration_network(x)
        return out[:, :, :args.height, :args.width]

    for _ in range(max(args.warmup, 1)):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    # HIP events cost ~10 us of stream time per instrumented launch (353 wm:: launches per step = 3 ms,
    # profiles/r01/README.md), so the timed region records only the scan kernel classes - the dominant
    # kernel `roofline` reports - and the other classes are measured in an extra untimed pass below.
    wm.ops.prof_enable(TIMED_PROF)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    prof_timed = wm.ops.prof_collect()
    prof_steps = {k: args.steps for k in TIMED_PROF}
    prof = {k: prof_timed[k] for k in TIMED_PROF}
    iso, iso_elapsed = {}, None
    if rank == 0:
        # untimed, one stream, back to back: every kernel class with the chip to itself (the timed region overlaps
        # kernels of different forwards, which stretches each launch)
        extra = max(2, min(args.steps, 5))
        torch.cuda.synchronize()
        wm.ops.prof_enable(True)
        t1 = time.perf_counter()
        for _ in range(extra):
            step(single=True)
        torch.cuda.synchronize()
        iso_elapsed = (time.perf_counter() - t1) / extra
        for k, v in wm.ops.prof_collect().items():
            iso[k] = (v[0] / extra, v[1] / extra)
            if k not in prof:
                prof[k], prof_steps[k] = v, extra
    wm.ops.prof_enable(False)
    concurrent = None
    if rank == 0 and world == 1 and args.concurrent > 1:
        cs = [torch.cuda.Stream(device) for _ in range(args.concurrent)]
        def cstep(i):
            with torch.no_grad(), torch.cuda.stream(cs[i % len(cs)]):
                net.restoration_network(x)
        for i in range(2 * len(cs)):
            cstep(i)
        torch.cuda.synchronize()
        kc = max(args.steps, 2 * len(cs))
        t1 = time.perf_counter()
        for i in range(kc):
            cstep(i)
        torch.cuda.synchronize()
        ce = time.perf_counter() - t1
        concurrent = {"streams": len(cs), "steps": kc, "images_per_s": kc / ce, "ms_per_image": 1e3 * ce / kc,
                      "note": "same forward, steps round-robin over the streams (that many images in flight), no HIP-event "
                              "instrumentation; serving throughput, not the contract's `value`"}
    op_boundary = scan_op_boundary(device, hp, wp) if rank == 0 else None
    def step_on_current_stream():           # capture needs the launches on the capturing stream: no stream switch
        with torch.no_grad():
            return net.restoration_network(x)[:, :, :args.height, :args.width]

    hip_graph = graph_replay(step_on_current_stream, args.steps, device) if rank == 0 and world == 1 and args.graph else None
    bf16 = bf16_autocast(net, x, args.steps) if rank == 0 and world == 1 and args.bf16 else None
    if world > 1:
        t = torch.tensor([elapsed], device="cpu" if share else device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)

    if rank == 0:
        # ---- per-kernel roofline table for the timed region (HIP-event durations from the library) ----
        l1, l2, l3 = (hp // 2) * (wp // 2), (hp // 4) * (wp // 4), (hp // 8) * (wp // 8)
        pos = scan_positions(hp, wp)                       # block-positions per image (14 LFSSBlocks)
        haar_b = 2 * 4 * 32 * hp * wp * (1 + 1 / 4 + 1 / 16)          # 2*e*B*C*H*W per level
        # algorithmic bytes per image of each hot-path kernel class
        algo = {
            "haar_analysis": haar_b, "haar_synthesis": haar_b,
            "ss2d_proj": pos * (256 + 4 * 144),            # read x (64 ch), write 4 records
            "ss2d_row_reduce": 2 * pos * (256 + 80), "ss2d_col_reduce": 2 * pos * (256 + 80),
            "ss2d_row_scan": 2 * pos * (256 + 144 + 256), "ss2d_col_scan": 2 * pos * (256 + 144 + 256),
            "selscan_chunk_scan": pos * SCAN_BYTES_PER_POS[16], "selscan_chunk_reduce": pos * 2304,
            "dwconv3x3": None,
        }
        table = {}
        for name, (n, ms) in prof.items():
            if not n:
                continue
            ks = prof_steps[name]
            ent = {"launches_per_step": n / ks, "ms_per_step": ms / ks,
                   "measured_in": "timed region" if name in TIMED_PROF else "untimed one-stream pass after it"}
            if algo.get(name):
                gbs = algo[name] * ks / (ms * 1e-3) / 1e9
                ent.update({"algorithmic_GB_per_step": algo[name] / 1e9, "achieved_GBps": gbs, "frac": gbs / HBM_PEAK_GBS})
            table[name] = ent
        hfe_only = ("conv3x3", "conv1x1", "skff")        # HFE-branch / plumbing kernels, not the hot path
        hot = [k for k in table if k != "dwconv3x3" and "frac" in table[k]]
        dom = max(hot, key=lambda k: table[k]["ms_per_step"])
        hot_ms = sum(table[k]["ms_per_step"] for k in table if k != "dwconv3x3" and k not in hfe_only)
        hfe_ms = sum(table[k]["ms_per_step"] for k in table if k in hfe_only)
        # transcendental ceiling of the scan kernels: KD*N = 4096 exp per block-position per pass
        exp_peak = 18.5e12                                  # v_exp_f32 lane-ops/s, tools/microbench.hip on MI355X
        scan_ms = sum(table[k]["ms_per_step"] for k in table if k.endswith(("_scan", "_reduce")))
        pmc = None
        try:                                               # measured HBM bytes per launch (rocprofv3 --pmc), if committed
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pmc = json.load(f)
        except Exception:
            pass
        roof = {
            "kernel": dom, "bound": "hbm", "achieved": table[dom]["achieved_GBps"], "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": table[dom]["frac"],
            # PMC FETCH_SIZE + WRITE_SIZE measured on the UHD level-1 launch of this kernel, scaled by the ratio
            # (measured / algorithmic) to the average launch of the timed region (launches differ by pyramid level)
            "traffic": ((pmc or {}).get(dom, {}).get("traffic_over_algorithmic") or 0) *
                       1e9 * table[dom]["algorithmic_GB_per_step"] / table[dom]["launches_per_step"] or None,
            "traffic_over_algorithmic": (pmc or {}).get(dom, {}).get("traffic_over_algorithmic"),
            "launches_per_step": table[dom]["launches_per_step"],
            "avg_launch_ms": table[dom]["ms_per_step"] / table[dom]["launches_per_step"],
            "algorithmic_bytes_per_launch_avg": 1e9 * table[dom]["algorithmic_GB_per_step"] / table[dom]["launches_per_step"],
            "note": "scan kernels are bound by v_exp_f32 issue (KD*N exp per position per pass), not by HBM: "
                    "see exp_frac; HBM fractions are reported for every hot-path kernel in roofline_table",
            "exp_frac_scan_kernels": (2 * 4096 * pos / (scan_ms * 1e-3) / exp_peak) if scan_ms else None,
            "hot_path_ms_per_step": hot_ms, "hfe_conv_skff_ms_per_step": hfe_ms,
        }
        if nstreams > 1 and dom in iso and iso[dom][1] > 0:
            # the same kernel class with the chip to itself (untimed one-stream pass): in the timed region kernels of
            # up to `streams` forwards overlap, which raises throughput and stretches every individual launch
            gbs_iso = algo[dom] / (iso[dom][1] * 1e-3) / 1e9
            roof["isolated"] = {"achieved": gbs_iso, "frac": gbs_iso / HBM_PEAK_GBS,
                                "avg_launch_ms": iso[dom][1] / iso[dom][0],
                                "ms_per_step_one_stream": 1e3 * iso_elapsed,
                                "images_per_s_one_stream": 1.0 / iso_elapsed,
                                "note": "one stream, back to back, events on every kernel class (~3 ms of event overhead per step)"}
        line = {
            "metric": "UHD (3840x2160) images/sec fwd", "value": world * args.steps / elapsed,
            "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"Wave-Mamba UHD-LL inference config (wf=32, n_l=[1,2,4], n_h=[1,1,2]), "
                                   f"1x3x{args.height}x{args.width} reflect-padded to {hp}x{wp}, seeded random "
                                   f"init, one image per GPU per step, replicas (no collective); steps round-robin "
                                   f"over {nstreams} HIP stream(s) = forwards in flight", "streams": nstreams},
            "roofline": roof, "roofline_table": table,
            "selscan_op_boundary": op_boundary, "hip_graph_replay": hip_graph, "bf16_autocast": bf16,
            "concurrent_forwards": concurrent, "cpu_baseline": cpu, "parity": parity,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
