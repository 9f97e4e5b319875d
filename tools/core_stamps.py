#!/usr/bin/env python3
"""Phase breakdown of the SS2D core kernels from in-kernel cycle stamps (library built with -DWM_CORE_STAMP=1):
   WAVEMAMBA_HIP_LIB=build/variants/stamp.so python tools/core_stamps.py [--level 1]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser(); ap.add_argument("--level", type=int, default=1); args = ap.parse_args()
dev = "cuda:0"
H, W = 2176 >> args.level, 3840 >> args.level
buf = torch.zeros(4096 * 16 * 12, dtype=torch.int64, device=dev)
os.environ["WM_CORE_STAMPS"] = str(buf.data_ptr())
import wave_mamba_amd as wm
D, N, R = 64, 16, 2
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.randn(1, D, H, W, device=dev, generator=g)
Wx = torch.randn(4, R + 2 * N, D, device=dev, generator=g) / 8
Wdt = torch.randn(4, D, R, device=dev, generator=g) * 0.7
bias = torch.randn(4, D, device=dev, generator=g) * 0.5 - 3.0
A_logs = torch.log(torch.arange(1, N + 1, device=dev, dtype=torch.float32)).repeat(4 * D, 1)
Ds = torch.ones(4 * D, device=dev)
for _ in range(2):
    wm.ops.ss2d_core(x, Wx, Wdt, bias, A_logs, Ds)
torch.cuda.synchronize()
buf.zero_()
wm.ops.ss2d_core(x, Wx, Wdt, bias, A_logs, Ds)
torch.cuda.synchronize()
t = buf.view(-1, 12).cpu()
t = t[t[:, 6] > 0]
names = ["stage+barrier", "fetch issue", "projection", "16 scan steps", "y store", "total"]
print(f"level {args.level} {H}x{W}: {len(t)} waves stamped (the last launch = chunk-scan kernel; shader cycles)")
if float(t[:, 5].double().sum()) > 0:          # -DWM_CORE_STAMP=1 build: per-phase totals
    for k in range(4):
        tk = t[t[:, 7] == k].double()
        if len(tk) == 0:
            continue
        per_tile = tk[:, :6] / tk[:, 6:7]
        m = per_tile.mean(0)
        print(f"  direction {k}: waves {len(tk)}, tiles/wave {tk[:,6].mean():.1f}; per tile: " +
              "  ".join(f"{n} {v:.1f}" for n, v in zip(names, m)) + f"  (sum of phases {float(m[:5].sum()):.1f})")
# timeline (wall_clock64: 100 MHz, chip-wide): workgroup entries and exits relative to the first entry
t0 = float(t[:, 8].min())
ent = (t[:, 8].double() - t0) / 100.0          # microseconds
ext = (t[:, 10].double() - t0) / 100.0
import collections
wg = slice(0, None, 16)
print(f"  kernel span (first entry -> last exit): {float(ext.max()):.1f} us")
hist = collections.Counter((ent[wg] / 50).floor().tolist())
print("  workgroup entries per 50 us bin:", [(int(50 * k), v) for k, v in sorted(hist.items())])
for kk in range(4):
    m = t[:, 7] == kk
    print(f"  direction {kk}: lifetime mean {float((ext[m] - ent[m]).mean()):.1f} us (min {float((ext[m] - ent[m]).min()):.1f}, max {float((ext[m] - ent[m]).max()):.1f}); "
          f"last exit {float(ext[m].max()):.1f} us")
# lifetime by wave index inside the workgroup (which waves are the fast ones: wave % 4 = SIMD, wave / 4 = launch order on it)
t_all = buf.view(-1, 16, 12).cpu()
for kk in (0, 1):
    sel = (t_all[:, 0, 6] > 0) & (t_all[:, 0, 7] == kk)
    if int(sel.sum()) == 0:
        continue
    w = t_all[sel].double()
    life = (w[:, :, 10] - w[:, :, 8]) / 100.0
    print(f"  direction {kk}: mean lifetime by wave index 0..15 (us): " + " ".join(f"{float(v):.0f}" for v in life.mean(0)))
    print(f"  direction {kk}: workgroup lifetime (max over its waves) mean {float(life.max(1).values.mean()):.1f} us, mean over waves {float(life.mean()):.1f} us")
