#!/usr/bin/env python3
"""bench.py - UHD (3840x2160) images/s of the Wave-Mamba forward on MI355X, with the HBM-roofline fraction of the
hot-path operator that takes the most time (SURVEY.md 8d bytes) and the CPU-oracle network timed beside it.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one forward of the shipped WaveMamba (inference_wavemamba.py:71-75 config, seeded random init -
checkpoints are not distributed) over one synthetic 1x3x2160x3840 image, reflect-padded to 2176x3840 exactly as the
reference's inference script does (:28-36), input already resident in HBM.  N > 1: one process per GPU, each an
independent replica on its own image (the path shards by image, no data-path collective) -> weak
scaling; value = N*K images / max-over-ranks time.

Definitions used in the JSON line (all from SURVEY.md section 8d, nothing else):
  * selective-scan op = the reduce / carry / scan launches of wm_ss2d_core_fwd (SS2D.forward_core :446-478);
    algorithmic bytes = 3584 B per scanned position (4*(3*KD + 2*K*N), KD = 256, K = 4, N = 16) - the reference call
    signature's operands, even though xs / dts / B / C never exist in HBM here;  `fused_512B` is the SEPARATE stretch
    definition (read x + write merged y = 512 B per position), never mixed into `frac`;
  * Haar: 2*e*B*C*H*W per level.
Prints ONE JSON line on rank 0.
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

SHIPPED = dict(in_chn=3, wf=32, n_l_blocks=[1, 2, 4], n_h_blocks=[1, 1, 2], ffn_scale=2.0)
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
SCAN_BYTES_PER_POS = {16: 3584, 32: 4096}   # 4*(3*KD + 2*K*N), KD = 256, K = 4
FUSED_BYTES_PER_POS = 512        # SURVEY.md 8d "fused SS2D-core (stretch, report separately, never mix)"
# kernel classes of the selective-scan op whose HIP events are recorded in the instrumented pass
CORE_CLASSES = ("ss2d_core_reduce", "selscan_carry", "ss2d_core_scan")
EXP_PEAK = 18.5e12               # v_exp_f32 lane-ops/s chip-wide, tools/microbench.hip on MI355X
# VALU issue cycles one wave (64 channels) spends per scan STEP in the innermost loops of wm::ss2d_core_kernel<16,16,1|3>, counted
# from the ISA of the shipped build by tools/isa_valu_count.py (profiles/r05/isa_valu_count.json): 4 cycles per VALU instruction
# (packed or not), 8 per transcendental.  Round 4's build: 289 / 325; round 5 (softplus in log2 units, constants folded): 279 / 315.
# The arithmetic itself (per step: 16 exponentials and 24 / 32 packed operations) is 224 / 256.
VALU_CYCLES_PER_STEP = {"reduce": 279, "scan": 315, "reduce_arithmetic_only": 224, "scan_arithmetic_only": 256}
SIMDS, SHADER_CLOCK_HZ = 1024, 2.4e9         # 256 CUs x 4 SIMDs; the clock the scan kernels sustain (tools/ubench_active_cus)
# The LFSSBlock kernels around the scan.  SURVEY 8d gives bytes for the wavelets and the scan
# only; for these the algorithmic bytes are what each kernel must move at the shipped width (C = 32, D = 64, fp32 planes):
#   lfss_in   reads tokens (4C) and writes x (4D) - and z (4D) unless the gate is recomputed downstream     = 384 (640) B / position
#   dwconv3x3 reads x and writes conv(x) (2 x 4D)                                          = 512
#   lfss_mid  reads the scan's `LFSS_MID_NY` y buffers and the tokens - and z unless it recomputes the gate from the tokens
#             (round 4, ops._RECOMPUTE_Z) -, writes tok1 and f (4 D NY [+ 4D] + 4C + 4C + 4D)  = 1536 (1792)
#   lfss_out  reads f (4D) and tok1 (4C), writes the block's output (4C)                   = 512
# The table quotes what the kernels the run actually used must move (never the larger figure for the smaller data flow).
LFSS_MID_NY = 4                                                        # the four directions' y planes, added by lfss_mid as it loads them


def lfss_bytes_per_pos():
    z = 0 if getattr(wm.ops, "_RECOMPUTE_Z", False) else 256
    return {"lfss_in": 128 + 256 + z, "dwconv3x3": 512, "lfss_mid": 256 * LFSS_MID_NY + z + 128 + 128 + 256, "lfss_out": 512}


LFSS_BYTES_PER_POS = lfss_bytes_per_pos()


MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 matrix peak of MI355X (MI355X_MICROARCH.md); never the 2:1-sparsity figure


def conv_workload(net, x):
    """Arithmetic of the dense convolutions of ONE forward (the HFE branch and the plumbing): every call of
    ops.conv2d / conv2d_gated / conv2d_ln is counted by shape while one forward runs.  -> {"3x3": (flop, bytes, calls), "1x1": ...};
    flop = 2 Cin Cout k^2 per output position (the fp32 convolution's; the kernels spend three bf16 products on each),
    bytes = 4 (Cin + Cout) per position (+ 4 Cout for a gate / residual operand)."""
    tally = {"3x3": [0.0, 0.0, 0], "1x1": [0.0, 0.0, 0]}
    o = wm.ops
    orig = (o.conv2d, o.conv2d_gated, o.conv2d_ln)

    def count(ks, pos, cin, cout, extra=0):
        t = tally["3x3" if ks == 3 else "1x1"]
        t[0] += 2.0 * cin * cout * ks * ks * pos; t[1] += 4.0 * (cin + cout + extra) * pos; t[2] += 1

    def conv2d(xx, weight, bias=None, x2=None, x2_index=None, gate=None, residual=None, dynamic_weight=False):
        pos = xx.shape[0] * xx.shape[2] * xx.shape[3]
        count(weight.shape[2], pos, weight.shape[1], weight.shape[0], weight.shape[0] * ((gate is not None) + (residual is not None)))
        return orig[0](xx, weight, bias, x2, x2_index, gate, residual, dynamic_weight)

    def conv2d_gated(xx, weight3, weight1, bias1=None, x2=None, x2_index=None):
        pos = xx.shape[0] * xx.shape[2] * xx.shape[3]
        count(3, pos, weight3.shape[1], weight3.shape[0])
        tally["3x3"][0] += 2.0 * weight1.shape[1] * weight1.shape[0] * pos        # the gating 1x1 rides in the same kernel
        return orig[1](xx, weight3, weight1, bias1, x2, x2_index)

    def conv2d_ln(xx, lw, lb, eps, weight, bias=None, residual=None):
        pos = xx.shape[0] * xx.shape[2] * xx.shape[3]
        count(1, pos, weight.shape[1], weight.shape[0], weight.shape[0] * (residual is not None))
        return orig[2](xx, lw, lb, eps, weight, bias, residual)

    o.conv2d, o.conv2d_gated, o.conv2d_ln = conv2d, conv2d_gated, conv2d_ln
    try:
        with torch.no_grad():
            net.restoration_network(x)
        torch.cuda.synchronize()
    finally:
        o.conv2d, o.conv2d_gated, o.conv2d_ln = orig
    return {k: tuple(v) for k, v in tally.items()}


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


def psnr_u8(a, b):
    """PSNR after the reference's uint8 quantisation (img_util.py:67-94, comput_psnr_ssim.py:434-438)."""
    qa, qb = (a.clamp(0, 1) * 255).round(), (b.clamp(0, 1) * 255).round()
    mse = float((qa - qb).pow(2).mean())
    return float("inf") if mse == 0 else float(20 * torch.log10(torch.tensor(255.0)) - 10 * torch.log10(torch.tensor(mse)))


# ------------------------------------------------------------------------------------------------
# rank plumbing (factored so that tests/test_bench_ranks.py can drive it under gloo on CPU)
# ------------------------------------------------------------------------------------------------
def rank_env(env=os.environ):
    return int(env.get("RANK", "0")), int(env.get("WORLD_SIZE", "1")), int(env.get("LOCAL_RANK", "0"))


def timed_steps(step, steps, warmup, sync, barrier):
    """W untimed warm-up steps, then EXACTLY `steps` steps bracketed by barrier + sync on both sides.
    -> seconds on this rank."""
    for _ in range(max(warmup, 0)):
        step()
    sync()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    barrier()
    return time.perf_counter() - t0


def max_over_ranks(seconds, world, device):
    if world <= 1:
        return float(seconds)
    t = torch.tensor([seconds], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t)


def whole_job_value(world, steps, units_per_step, seconds):
    """units all ranks processed / max-over-ranks time (weak scaling: every rank does `steps` steps)."""
    return world * steps * units_per_step / seconds


def device_identity(device):
    """What makes this rank's GPU THIS GPU: name, uuid (when the runtime reports one), PCI bus id, visible-device settings."""
    pr = torch.cuda.get_device_properties(device)
    ident = {"index": device.index, "name": pr.name, "uuid": str(getattr(pr, "uuid", "")) or None,
             "pci": f"{getattr(pr, 'pci_domain_id', 0):04x}:{getattr(pr, 'pci_bus_id', -1):02x}:{getattr(pr, 'pci_device_id', -1):02x}",
             "compute_units": pr.multi_processor_count, "hbm_GiB": round(pr.total_memory / 2 ** 30, 1),
             "visible_devices": os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES")),
             "pid": os.getpid()}
    return ident


def gather_identities(ident, world):
    """Every rank's device_identity() on every rank (all_gather_object over the job's process group)."""
    if world <= 1 or not (dist.is_available() and dist.is_initialized()):
        return [ident]
    out = [None] * world
    dist.all_gather_object(out, ident)
    return out


def distinct_devices(idents):
    """N ranks saw N different GPUs?  Keyed by uuid when reported, else by PCI address (+ index)."""
    keys = [(i.get("uuid") or (i.get("pci"), i.get("index"))) for i in idents]
    return len(set(keys)) == len(keys)


def valu_floor(pos, core_ms, iso_ms):
    """The op's VALU-issue floor: wave-steps x issue cycles per step of BOTH passes / (SIMDs x clock).
    `valu_floor_ms` takes the instruction stream as compiled (ISA count), `valu_arithmetic_floor_ms` only the recurrence's own
    arithmetic; the fractions say how much of the measured op time those floors explain (in the step as timed / alone)."""
    wave_steps = 4 * pos                                   # four directions, one wave of 64 channels per position
    per = VALU_CYCLES_PER_STEP
    f = lambda cyc: cyc * wave_steps / (SIMDS * SHADER_CLOCK_HZ) * 1e3
    floor, arith = f(per["reduce"] + per["scan"]), f(per["reduce_arithmetic_only"] + per["scan_arithmetic_only"])
    return {"valu_cycles_per_step": per, "valu_floor_ms": floor, "valu_arithmetic_floor_ms": arith,
            "valu_floor_frac": floor / core_ms if core_ms else None,
            "valu_floor_frac_isolated": floor / iso_ms if iso_ms else None,
            "valu_floor_source": "tools/isa_valu_count.py on the shipped build: innermost step loops of ss2d_core_kernel<16,16,1|3> "
                                 "(projection, staging, y stores, prologues, tails and the carry are not in the floor)"}


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_command(nproc, argv, port=None, script=None):
    """The command line `python bench.py --gpus N` turns into when no launcher set WORLD_SIZE."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()),
            script or os.path.abspath(__file__)] + list(argv)


def self_launch(nproc, argv):
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on these hosts (RCCL needs it)
    return subprocess.call(launch_command(nproc, argv), env=env)


def image_seed(rank):
    return 1234 + rank                                 # G(1234) on rank 0; every replica its own image


# ------------------------------------------------------------------------------------------------
# CPU baseline: the same network with the C/OpenMP oracle as hot-path backend, on host cores
# ------------------------------------------------------------------------------------------------
def cpu_baseline(img_padded, forwards, budget_s):
    """1 warm-up + `forwards` timed forwards of the FULL padded UHD image on the host.
    -> (record, output of the last forward).  If the warm-up alone exceeds `budget_s` the timed count drops to 1 and
    the record says so."""
    from oracle import oracle
    from oracle import backend as oracle_backend
    cores = oracle.usable_cpus(cap=1 << 20)            # affinity mask and cgroup quota: what this process may really use
    torch.set_num_threads(cores)
    oracle.set_num_threads(cores)
    net = build_model("cpu")
    times = []
    with oracle_backend.ops_backend(oracle), torch.no_grad():
        t0 = time.perf_counter()
        y = net.restoration_network(img_padded)        # warm-up
        warm = time.perf_counter() - t0
        n = forwards if warm * (forwards + 1) <= budget_s else 1
        for _ in range(n):
            t0 = time.perf_counter()
            y = net.restoration_network(img_padded)
            times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    h, w = img_padded.shape[-2:]
    return {
        "value": 1.0 / med, "unit": "images/s", "cores": cores, "kind": "port",
        "host_cpus": os.cpu_count(), "torch_threads": torch.get_num_threads(), "oracle_omp_threads": oracle.num_threads(),
        "seconds_per_image": times, "warmup_seconds": warm,
        "sample": f"the full workload: 1 warm-up + {len(times)} timed forwards of the padded 1x3x{h}x{w} image (median "
                  f"{med:.2f} s), same network code with the C/OpenMP oracle (oracle/wavemamba_oracle.c) as hot-path "
                  f"backend + PyTorch-CPU for the rest; threads = host cores usable by this process (affinity and "
                  f"cgroup quota) = {cores} of os.cpu_count() = {os.cpu_count()}",
    }, y


def scan_op_boundary(device, hp, wp, iters=3):
    """The drop-in selective_scan_fn (reference call signature, 3584 B/position) on the UHD level-1 shape (B=1, KD=256,
    L=hp*wp/4): HBM fraction of the whole op."""
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
            "reduce_ms": r, "carry_ms": c, "chunk_scan_ms": s,
            "whole_op_frac": nbytes / ((r + c + s) * 1e-3) / 1e9 / HBM_PEAK_GBS}


def graph_replay(step, steps, device):
    """The same step captured once into a HIP graph and replayed (optional leg)."""
    try:
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()                                              # warm the private-pool allocations
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        # (a process group's watchdog thread polls events while this thread captures: thread-local capture mode, trainer.py)
        mode = {"capture_error_mode": "thread_local"} if dist.is_initialized() else {}
        torch.cuda.synchronize()
        with torch.cuda.graph(g, **mode):
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


SCAN_BWD_BYTES_PER_POS = 6144    # 4*(5*KD + 4*K*N): reads u, delta, dy, B, C + writes du, ddelta, dB, dC


def train_leg(device, steps, with_cpu_loss, batch=8, size=512):
    """BASELINE config 3 on ONE GPU (optional leg, rank 0): `steps` optimize_parameters() of the shipped config on a
    synthetic batch of 8 x 3 x 512 x 512 pairs (femasr_model.py:157-185: forward, L1 + 0.1 FFT-L1, backward, AdamW) -
    images/s, the backward of the selective-scan op as a fraction of the HBM roofline on SURVEY.md 8d's 6144 B per
    position, and the first step's loss against the same network on the host with the CPU oracle as hot-path backend."""
    torch.manual_seed(0)
    net = wm.WaveMamba(**SHIPPED).train().to(device)
    opt = wm.trainer.make_optimizer(net)
    g = torch.Generator().manual_seed(image_seed(0))
    lq_cpu, gt_cpu = torch.rand(batch, 3, size, size, generator=g), torch.rand(batch, 3, size, size, generator=g)
    lq, gt = lq_cpu.to(device), gt_cpu.to(device)
    loss_parity = cpu_train = None
    with torch.no_grad():
        l_pix, l_fft = wm.trainer.losses(net(lq), gt)
    first = (float(l_pix), float(l_fft))
    if with_cpu_loss:
        from oracle import oracle
        from oracle import backend as oracle_backend
        torch.manual_seed(0)
        net_cpu = wm.WaveMamba(**SHIPPED).train()
        with oracle_backend.ops_backend(oracle), torch.no_grad():
            c_pix, c_fft = wm.trainer.losses(net_cpu(lq_cpu), gt_cpu)
        loss_parity = {"gpu": first, "cpu_oracle_network": (float(c_pix), float(c_fft)),
                       "rel_diff": max(abs(first[0] - float(c_pix)) / float(c_pix), abs(first[1] - float(c_fft)) / float(c_fft)),
                       "bar": 1e-6}
        # the number beside the GPU's: the same optimize_parameters() on the host cores, on a bounded sample (one pair of the batch)
        cores = oracle.usable_cpus(cap=1 << 20)
        torch.set_num_threads(cores)
        oracle.set_num_threads(cores)
        opt_cpu = wm.trainer.make_optimizer(net_cpu)
        times = []
        with oracle_backend.ops_backend(oracle):
            for _ in range(2):                                              # 1 warm-up + 1 timed step
                t0 = time.perf_counter()
                wm.trainer.train_step(net_cpu, opt_cpu, lq_cpu[:1], gt_cpu[:1])
                times.append(time.perf_counter() - t0)
                if times[0] > 25.0:                                         # slow host: the first step is the sample
                    break
        cpu_train = {"value": 1.0 / times[-1], "unit": "images/s", "cores": cores, "kind": "port", "seconds_per_step": times,
                     "sample": f"one optimize_parameters() on 1 of the {batch} pairs (1x3x{size}x{size}), the last of {len(times)} step(s): the same "
                               f"network and trainer code with the C/OpenMP oracle as hot-path backend (forward and backward) + "
                               f"PyTorch-CPU for the rest, {cores} threads"}
        del net_cpu, opt_cpu
    state = {}

    def step():
        state["losses"] = wm.trainer.train_step(net, opt, lq, gt, as_float=False)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    wm.ops.prof_enable(("selscan_bwd", "ss2d_core_reduce", "ss2d_core_scan", "selscan_carry"))
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    prof = wm.ops.prof_collect()
    wm.ops.prof_enable(False)
    peak_eager = torch.cuda.max_memory_allocated(device) / 2 ** 30      # before the graphed leg's second network + graph pool
    graphed = None
    try:                                                            # the same step captured into a HIP graph: no host work per step
        torch.manual_seed(0)
        net_g = wm.WaveMamba(**SHIPPED).train().to(device)
        torch.cuda.reset_peak_memory_stats(device)
        gstep = wm.trainer.GraphedTrainStep(net_g, wm.trainer.make_optimizer(net_g, capturable=True), lq, gt)
        gstep(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            gstep()
        torch.cuda.synchronize()
        dg = time.perf_counter() - t0
        graphed = {"images_per_s": steps * batch / dg, "ms_per_step": 1e3 * dg / steps,
                   "peak_mem_GB_with_the_eager_leg_resident": torch.cuda.max_memory_allocated(device) / 2 ** 30,
                   "note": "trainer.GraphedTrainStep: forward, losses, backward and AdamW captured once and replayed"}
        del net_g, gstep
    except Exception as e:                                          # best-effort leg, never fatal
        graphed = {"error": f"{type(e).__name__}: {e}"[:300]}
    pos = batch * scan_positions(size, size)
    bwd_ms = prof["selscan_bwd"][1] / steps
    fwd_ms = sum(prof[k][1] for k in CORE_CLASSES) / steps
    res = {"workload": f"BASELINE config 3 on one GPU: batch {batch} x 3x{size}x{size} synthetic pairs, shipped config, "
                       f"L1 + 0.1 FFT-L1, AdamW(5e-4, wd 1e-3, betas (0.9, 0.99)); no DDP at N = 1",
           "images_per_s": steps * batch / dt, "ms_per_step": 1e3 * dt / steps, "steps": steps,
           "scan_positions_per_step": pos,
           "selective_scan_backward": {
               "kernels": "wm_ss2d_core_bwd (transposes, projection, chunked adjoint scan, projection backward, reductions)",
               "ms_per_step": bwd_ms, "algorithmic_bytes_per_position": SCAN_BWD_BYTES_PER_POS,
               "algorithmic_GB_per_step": SCAN_BWD_BYTES_PER_POS * pos / 1e9,
               "frac": SCAN_BWD_BYTES_PER_POS * pos / (bwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if bwd_ms else None},
           "selective_scan_forward": {"ms_per_step": fwd_ms,
                                      "frac": SCAN_BYTES_PER_POS[16] * pos / (fwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if fwd_ms else None},
           "losses_after_steps": wm.trainer.loss_values(state["losses"]), "first_step_loss_parity": loss_parity,
           "cpu_baseline": cpu_train, "hip_graph_replay": graphed,
           "peak_mem_GB": peak_eager}
    del net, opt
    torch.cuda.empty_cache()
    return res


def ddp_train_leg(device, rank, world, steps, reduce_device, batch=8, size=512):
    """BASELINE config 3 as worded (N > 1, every rank): DistributedDataParallel over the ranks (base_model.py:111-114), batch 8
    synthetic 512 x 512 pairs per GPU, `steps` optimize_parameters() between barriers, max over ranks -> whole-job training
    images/s (weak scaling).  The same steps under `no_sync()` (no gradient all-reduce) give the exposed cost of the one
    exchange step this path has: the 6.05 MB gradient all-reduce over RCCL / xGMI."""
    torch.manual_seed(0)
    net = wm.WaveMamba(**SHIPPED).train().to(device)
    model = wm.trainer.wrap_ddp(net, device, force=(world == 1))       # world 1: the one-rank RCCL self-test
    opt = wm.trainer.make_optimizer(model)
    g = torch.Generator().manual_seed(image_seed(rank))
    lq, gt = torch.rand(batch, 3, size, size, generator=g).to(device), torch.rand(batch, 3, size, size, generator=g).to(device)
    state = {}

    def step():
        state["losses"] = wm.trainer.train_step(model, opt, lq, gt, as_float=False)

    def step_nosync():
        with model.no_sync():
            wm.trainer.train_step(model, opt, lq, gt, as_float=False)
    sync, barrier = torch.cuda.synchronize, dist.barrier
    t_ddp = max_over_ranks(timed_steps(step, steps, 2, sync, barrier), world, reduce_device)
    losses = wm.trainer.loss_values(state["losses"])
    t_local = None
    if hasattr(model, "no_sync"):
        t_local = max_over_ranks(timed_steps(step_nosync, steps, 1, sync, barrier), world, reduce_device)
    nparam = sum(p.numel() for p in net.parameters())
    # the same data-parallel step with the host out of it (trainer.GraphedDDPTrainStep): forward + backward + flat gradient buffer
    # replayed from a HIP graph, ONE all-reduce, AdamW replayed.  Collective = one call, so every rank must take this leg or none:
    # a rank that fails says so in the all-reduced flag and all ranks skip the timing
    graphed = None
    try:
        del model, opt
        torch.cuda.empty_cache()
        # 'split' (the all-reduce is an ordinary call between two replays) unless asked otherwise: a collective INSIDE a HIP graph
        # has only ever run on a one-rank communicator from the build sessions (tests/test_rccl_gpu.py)
        mode = "captured" if dist.get_backend() == "nccl" and os.environ.get("WM_BENCH_DDP_GRAPH") == "captured" else "split"
        err = None
        try:
            torch.manual_seed(0)
            net_g = wm.WaveMamba(**SHIPPED).train().to(device)      # a fresh module: no reducer hooks on its parameters
            gstep = wm.trainer.GraphedDDPTrainStep(net_g, wm.trainer.make_optimizer(net_g, capturable=True), lq, gt, collective=mode)
        except Exception as e:
            err = f"{type(e).__name__}: {e}"[:300]
        bad = torch.tensor([0.0 if err is None else 1.0], device=reduce_device)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if float(bad) == 0.0:
            t_g = max_over_ranks(timed_steps(lambda: gstep(), steps, 1, sync, barrier), world, reduce_device)
            graphed = {"images_per_s": whole_job_value(world, steps, batch, t_g), "ms_per_step": 1e3 * t_g / steps,
                       "collective": mode, "losses_mean_over_ranks": wm.trainer.loss_values(gstep.losses),
                       "note": "trainer.GraphedDDPTrainStep: the bare module's forward, losses and backward + gradients into one flat "
                               "buffer replayed from a HIP graph, one all-reduce of the buffer (inside the graph when 'captured'), "
                               "AdamW on views of the buffer replayed"}
        else:
            graphed = {"error": err or "another rank failed to capture"}
    except Exception as e:
        graphed = {"error": f"{type(e).__name__}: {e}"[:300]}
    res = {"workload": f"BASELINE config 3: DDP over {world} ranks, batch {batch} x 3x{size}x{size} synthetic pairs per GPU, "
                       f"shipped config, L1 + 0.1 FFT-L1, AdamW; one gradient all-reduce of {4 * nparam / 1e6:.2f} MB per step "
                       f"+ the 2-scalar loss reduce (base_model.py:392)",
           "images_per_s": whole_job_value(world, steps, batch, t_ddp), "ms_per_step": 1e3 * t_ddp / steps, "steps": steps,
           "ms_per_step_without_allreduce": None if t_local is None else 1e3 * t_local / steps,
           "exposed_allreduce_ms_per_step": None if t_local is None else 1e3 * (t_ddp - t_local) / steps,
           "scaling": "weak", "losses_rank0_mean": losses, "hip_graph_replay": graphed,
           "peak_mem_GB": torch.cuda.max_memory_allocated(device) / 2 ** 30}
    del net
    torch.cuda.empty_cache()
    return res


def load_pmc(hp, wp):
    """HBM bytes of the selective-scan op measured with rocprofv3 --pmc (tools/pmc_core.sh -> tools/pmc_traffic.py ->
    profiles/pmc_traffic.json): per core call at each pyramid level, FETCH_SIZE corrected as
    MI355X_MICROARCH.md prescribes (x2 for 16-byte-per-lane reads, the factor checked on a float4 copy in the same
    session).  -> (bytes per step or None when the file does not cover this workload, the file's content)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pmc = json.load(f)
        if pmc.get("build_id") != wm._lib.build_id():       # counters of another binary are not this binary's traffic
            return None, {"source": f"profiles/pmc_traffic.json describes build {pmc.get('build_id')}, the loaded library "
                                    f"is {wm._lib.build_id()}: not quoted"}
        lv = pmc["ss2d_core"]["levels"]
        calls = {1: 2, 2: 4, 3: 8}
        tot = 0.0
        for lvl, n in calls.items():
            e = lv[str(lvl)]
            if (e["H"], e["W"]) != (hp >> lvl, wp >> lvl):
                return None, pmc
            tot += n * e["hbm_bytes_per_call"]
        return tot, pmc
    except Exception:
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--concurrent", type=int, default=4,
                    help="extra untimed leg at N = 1: throughput with this many forwards in flight on separate HIP "
                         "streams, no event instrumentation (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--timed-only", action="store_true",
                    help="profiling runs: skip the untimed single-stream pass, the op-boundary leg and the concurrent leg, so "
                         "that a kernel trace of the command holds warm-up and timed steps only")
    ap.add_argument("--cpu-forwards", type=int, default=3, help="timed CPU forwards of the full UHD image (after 1 warm-up)")
    ap.add_argument("--cpu-budget", type=float, default=240.0,
                    help="seconds the CPU leg may take; if (1 + cpu-forwards) x warm-up time exceeds it, one timed forward")
    ap.add_argument("--graph", action="store_true", help="also time the step replayed from a HIP graph")
    ap.add_argument("--no-train", action="store_true",
                    help="skip the training leg (BASELINE config 3: batch 8 x 512 x 512 per GPU; N = 1: rank 0 alone, "
                         "N > 1: DistributedDataParallel over all ranks)")
    ap.add_argument("--train-steps", type=int, default=5)
    ap.add_argument("--no-bf16", action="store_true",
                    help="skip the bf16-storage leg (BASELINE config 2 as worded: bf16 planes between the kernels; rank 0, N = 1)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without an outer launcher: start the N ranks ourselves, exactly as the driver would
        # (one process per GPU under torch.distributed.run, rendezvous on 127.0.0.1), and hand its exit code back.
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    rank, world, local_rank = rank_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path is HIP-only (no CPU fallback)")
    # self-test hook for one-GPU boxes (never set by the driver): WM_BENCH_SHARE_GPU=1 puts every rank on cuda:0 over gloo,
    # to run the real N > 1 control flow (env parsing, barriers, max-over-ranks reduction, rank-0-only legs) on hardware;
    # the throughput of such a run means nothing (the ranks share one GPU)
    share = os.environ.get("WM_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # second self-test hook (never set by the driver): WM_BENCH_RCCL_SELFTEST=1 at N = 1 under torch.distributed.run creates the
    # RCCL communicator anyway (one rank) and runs the DDP training leg over it next to the plain one - the distributed calls
    # of the N > 1 path on real hardware with the only GPU a build session can reach
    rccl_selftest = world == 1 and os.environ.get("WM_BENCH_RCCL_SELFTEST") == "1" and "MASTER_ADDR" in os.environ
    if world > 1 or rccl_selftest:
        if share:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)      # RCCL on ROCm
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # who is where: every rank's GPU identity gathered over the job's own process group, so the line
    # itself shows that RCCL saw N ranks on N distinct devices; replicas on one shared device are an error, not a number
    idents = gather_identities(device_identity(device), world if dist.is_initialized() else 1)
    if world > 1 and not share and not distinct_devices(idents):
        raise SystemExit(f"bench.py: {world} ranks but the devices are not distinct: {idents}")
    ranks_info = {"world_size": dist.get_world_size() if dist.is_initialized() else 1, "launcher_world_size": world,
                  "backend": (dist.get_backend() if dist.is_initialized() else None),
                  "rccl_version": (".".join(str(v) for v in torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else None),
                  "distinct_devices": distinct_devices(idents), "devices": idents,
                  "torch": torch.__version__, "hip": torch.version.hip}

    img = torch.rand(1, 3, args.height, args.width, generator=torch.Generator().manual_seed(image_seed(rank)))
    x_cpu = pad_to(img)
    hp, wp = x_cpu.shape[-2:]

    # CPU baseline first (rank 0, N = 1 only), before the GPU pass: full UHD forwards
    cpu, y_cpu = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, y_cpu = cpu_baseline(x_cpu, args.cpu_forwards, args.cpu_budget)

    net = build_model(device)
    x = x_cpu.to(device)

    parity = None
    if y_cpu is not None:   # parity of the HIP path vs the CPU-oracle network on the very same UHD image
        with torch.no_grad():
            yg = net.restoration_network(x).cpu()
        tgt = torch.rand(y_cpu.shape, generator=torch.Generator().manual_seed(4321))
        crop = lambda t: t[:, :, :args.height, :args.width]
        parity = {"workload": f"the timed workload itself (1x3x{hp}x{wp}, cropped to {args.height}x{args.width} for PSNR)",
                  "rel_l2_vs_cpu_oracle": float((yg - y_cpu).norm() / y_cpu.norm()),
                  "max_abs_over_max_abs": float((yg - y_cpu).abs().max() / y_cpu.abs().max()),
                  "abs_dpsnr_db": abs(psnr_u8(crop(yg), crop(tgt)) - psnr_u8(crop(y_cpu), crop(tgt))),
                  "bars": {"rel_l2": 1e-4, "abs_dpsnr_db": 1e-3}}
        del yg, y_cpu

    def step():
        with torch.no_grad():
            out = net.restoration_network(x)
        return out[:, :, :args.height, :args.width]

    sync = torch.cuda.synchronize
    barrier = dist.barrier if (world > 1 or rccl_selftest) else (lambda: None)
    # `value` comes from an UN-instrumented pass: warm-up, then exactly K steps between barriers.  The roofline numbers come
    # from a second pass of K steps with HIP events around the launches of the selective-scan op (~10 us of stream time per
    # instrumented launch, 42 per step) - its wall time is reported next to it, never as `value`.
    elapsed = timed_steps(step, args.steps, max(args.warmup, 1), sync, barrier)
    wm.ops.prof_enable(CORE_CLASSES)
    elapsed_instr = timed_steps(step, args.steps, 0, sync, barrier)
    prof = {k: v for k, v in wm.ops.prof_collect().items() if k in CORE_CLASSES}
    prof_steps = {k: args.steps for k in prof}
    iso = {}
    if rank == 0 and not args.timed_only:
        # the untimed pass runs the single-stream order: every launch alone on the GPU (in the timed region the down
        # path's high-frequency branches run on side streams under the main stream's kernels, which lengthens both)
        extra = max(2, min(args.steps, 5))
        unet = net.restoration_network
        two = getattr(unet, "two_streams", False)
        unet.two_streams = False
        wm.ops.prof_enable(True)
        for _ in range(extra):
            step()
        torch.cuda.synchronize()
        for k, v in wm.ops.prof_collect().items():
            if k not in prof:
                prof[k], prof_steps[k] = v, extra
            elif k in CORE_CLASSES and v[0]:
                iso[k] = v[1] / extra
        unet.two_streams = two
    wm.ops.prof_enable(False)

    concurrent = None
    if rank == 0 and world == 1 and args.concurrent > 1 and not args.timed_only:
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
    op_boundary = scan_op_boundary(device, hp, wp) if rank == 0 and not args.timed_only else None
    hip_graph = graph_replay(step, args.steps, device) if rank == 0 and world == 1 and args.graph else None
    bf16 = None
    if rank == 0 and world == 1 and not args.no_bf16 and not args.timed_only:
        try:
            from wave_mamba_amd import inference
            bf16 = inference.bench_bf16_storage(net, x, args.steps)
        except Exception as e:
            bf16 = {"error": f"{type(e).__name__}: {e}"[:300]}
    train = None
    if rank == 0 and world == 1 and not args.no_train and not args.timed_only:
        try:
            train = train_leg(device, args.train_steps, with_cpu_loss=not args.no_cpu_baseline)
        except Exception as e:
            train = {"error": f"{type(e).__name__}: {e}"[:300]}
    ddp = None
    if (world > 1 or rccl_selftest) and not args.no_train and not args.timed_only:
        try:
            ddp = ddp_train_leg(device, rank, world, args.train_steps, "cpu" if share else device)
        except Exception as e:
            ddp = {"error": f"{type(e).__name__}: {e}"[:300]}
    elapsed = max_over_ranks(elapsed, world, "cpu" if share else device)

    if rank == 0:
        pos = scan_positions(hp, wp)                       # positions scanned per image (14 LFSSBlocks)
        haar_b = 2 * 4 * 32 * hp * wp * (1 + 1 / 4 + 1 / 16)          # 2*e*B*C*H*W per level
        table = {}
        for name, (n, ms) in prof.items():
            if n:
                ks = prof_steps[name]
                table[name] = {"launches_per_step": n / ks, "ms_per_step": ms / ks,
                               "measured_in": "instrumented pass (multi-stream, as timed)" if name in CORE_CLASSES
                               else "single-stream pass after it"}
        for name in ("haar_analysis", "haar_synthesis"):
            if name in table:
                gbs = haar_b / (table[name]["ms_per_step"] * 1e-3) / 1e9
                table[name].update({"algorithmic_GB_per_step": haar_b / 1e9, "achieved_GBps": gbs, "frac": gbs / HBM_PEAK_GBS})
        for name, bpp in LFSS_BYTES_PER_POS.items():
            if name in table:
                gb = bpp * pos / 1e9                          # one launch per LFSSBlock over its L positions: `pos` in total
                gbs = gb / (table[name]["ms_per_step"] * 1e-3)
                table[name].update({"algorithmic_bytes_per_position": bpp, "algorithmic_GB_per_step": gb,
                                    "achieved_GBps": gbs, "frac": gbs / HBM_PEAK_GBS})
        # the dense convolutions: 3x3 against the bf16 matrix peak (three products per fp32 product), 1x1 against HBM
        try:
            cw = conv_workload(net, x)
            if "conv3x3" in table and cw["3x3"][2]:
                fl, _, nc = cw["3x3"]
                tf = fl / (table["conv3x3"]["ms_per_step"] * 1e-3) / 1e12
                table["conv3x3"].update({"bound": "mfma", "algorithmic_TFLOP_per_step": fl / 1e12, "calls_counted": nc,
                                         "achieved_TFLOPs_fp32_equivalent": tf, "bf16_products_per_fp32_product": 3,
                                         "peak_TFLOPs_bf16_dense": MFMA_BF16_PEAK_TFLOPS, "frac": 3.0 * tf / MFMA_BF16_PEAK_TFLOPS})
            if "conv1x1" in table and cw["1x1"][2]:
                _, by, nc = cw["1x1"]
                gbs = by / (table["conv1x1"]["ms_per_step"] * 1e-3) / 1e9
                table["conv1x1"].update({"bound": "hbm", "algorithmic_GB_per_step": by / 1e9, "calls_counted": nc,
                                         "achieved_GBps": gbs, "frac": gbs / HBM_PEAK_GBS})
        except Exception as e:                             # a counting pass must never cost the line
            table.setdefault("conv3x3", {})["workload_error"] = f"{type(e).__name__}: {e}"[:200]
        core_ms = sum(table[k]["ms_per_step"] for k in CORE_CLASSES if k in table)
        calls = table.get("ss2d_core_scan", {}).get("launches_per_step", 0)
        scan_bytes = SCAN_BYTES_PER_POS[16] * pos
        achieved = scan_bytes / (core_ms * 1e-3) / 1e9 if core_ms else None
        traffic, pmc = load_pmc(hp, wp)
        dom = max((k for k in CORE_CLASSES if k in table), key=lambda k: table[k]["ms_per_step"], default=None)
        roof = {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS if achieved else None,
            "traffic": traffic,
            "kernel": "selective-scan op of SS2D.forward_core = the reduce + carry + scan launches of wm_ss2d_core_fwd "
                      "(wm::ss2d_core_kernel<16,16,1|3>, wm::selscan_carry_kernel): the hot-path operator with the most time",
            "definition": "SURVEY.md 8d: 3584 B per scanned position (reference call signature) x positions per step / "
                          "summed HIP-event duration of the op's launches in the instrumented pass (K steps right after the "
                          "timed region, same step, same streams)",
            "instrumented_pass_ms_per_step": 1e3 * elapsed_instr / args.steps,
            "algorithmic_bytes_per_position": SCAN_BYTES_PER_POS[16], "positions_per_step": pos,
            "algorithmic_GB_per_step": scan_bytes / 1e9, "ms_per_step": core_ms, "op_calls_per_step": calls,
            "per_call_avg": {"algorithmic_GB": scan_bytes / 1e9 / calls if calls else None,
                             "ms": core_ms / calls if calls else None},
            "dominant_kernel": None if dom is None else {
                "class": dom, "launches_per_step": table[dom]["launches_per_step"], "ms_per_step": table[dom]["ms_per_step"],
                "avg_launch_ms": table[dom]["ms_per_step"] / table[dom]["launches_per_step"]},
            "traffic_over_algorithmic": traffic / scan_bytes if traffic else None,
            "traffic_source": None if pmc is None else pmc.get("source"),
            "fused_512B": {"note": "SURVEY.md 8d stretch definition, reported separately, never mixed into `frac`",
                           "algorithmic_GB_per_step": FUSED_BYTES_PER_POS * pos / 1e9,
                           "frac": FUSED_BYTES_PER_POS * pos / (core_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if core_ms else None,
                           "traffic_over_algorithmic": traffic / (FUSED_BYTES_PER_POS * pos) if traffic else None},
            "isolated": None if len(iso) != len(CORE_CLASSES) else {
                "note": "the same launches with nothing else on the GPU (untimed single-stream pass after the timed region); "
                        "`achieved` / `frac` above are the contract's: durations in the step as it is timed, where side-stream "
                        "kernels of the high-frequency branch share the compute units with them",
                "ms_per_step": sum(iso.values()),
                "achieved": scan_bytes / (sum(iso.values()) * 1e-3) / 1e9,
                "frac": scan_bytes / (sum(iso.values()) * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "secondary_ceilings": {
                "note": "the op is bound by VALU issue, not by HBM: KD*N = 4096 v_exp_f32 per position in each of the two "
                        "passes (chunk-reduce, chunk-scan) plus four packed fp32 operations per state-step",
                "exp_frac": (2 * 4096 * pos / (core_ms * 1e-3) / EXP_PEAK) if core_ms else None,
                **valu_floor(pos, core_ms, sum(iso.values()) if len(iso) == len(CORE_CLASSES) else None)},
        }
        hot_names = ("haar_analysis", "haar_synthesis", "lfss_in", "lfss_mid", "lfss_out", "dwconv3x3") + CORE_CLASSES
        # SURVEY.md 8d "for the sum": DWT + IWT + scans = 2.81 + 2.81 + 26.20 GB per UHD image over the time of EVERY
        # hot-path kernel (wavelets, LFSSBlock glue, depth-wise conv, the scan op), each alone on the GPU
        hot_iso_ms = sum((iso[k] if k in iso else table[k]["ms_per_step"]) for k in table if k in hot_names)
        hot_bytes = 2 * haar_b + scan_bytes
        hot_sum = {"definition": "SURVEY.md 8d hot-path sum: DWT + IWT (2*e*B*C*H*W per level each) + scans (3584 B / position), "
                                 "over the summed duration of every hot-path kernel class (haar, lfss_in / mid / out, depth-wise "
                                 "conv + SiLU, scan reduce / carry / scan), single-stream pass",
                   "algorithmic_GB_per_step": hot_bytes / 1e9, "ms_per_step": hot_iso_ms,
                   "achieved_GBps": hot_bytes / (hot_iso_ms * 1e-3) / 1e9 if hot_iso_ms else None,
                   "frac": hot_bytes / (hot_iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if hot_iso_ms else None,
                   "ceiling_images_per_s": HBM_PEAK_GBS * 1e9 / hot_bytes}
        line = {
            "metric": "UHD (3840x2160) images/sec fwd", "value": whole_job_value(world, args.steps, 1, elapsed),
            "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic" + (" (WM_BENCH_SHARE_GPU self-test: ranks share one GPU)" if share else ""),
            "config": {"workload": f"Wave-Mamba UHD-LL inference config (wf=32, n_l=[1,2,4], n_h=[1,1,2]), "
                                   f"1x3x{args.height}x{args.width} reflect-padded to {hp}x{wp}, seeded random "
                                   f"init, one image per GPU per step, replicas (no collective); one forward at a time, "
                                   f"its down-path high-frequency branches on side streams: "
                                   f"{bool(getattr(net.restoration_network, 'two_streams', False))}"},
            "ranks": ranks_info,
            "roofline": roof, "cpu_baseline": cpu, "parity": parity,
            "roofline_table": table,
            "hot_path_ms_per_step": sum(table[k]["ms_per_step"] for k in table if k in hot_names),
            "hot_path_sum": hot_sum,
            "selscan_op_boundary": op_boundary, "hip_graph_replay": hip_graph, "bf16_storage": bf16,
            "concurrent_forwards": concurrent, "training_config3_one_gpu": train, "training_config3_ddp": ddp,
        }
        print(json.dumps(line))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
