#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the rocprofv3 --pmc passes of tools/pmc_core.sh: HBM bytes of the selective-scan op
(reduce + carry + scan launches of wm_ss2d_core_fwd) per call at each UHD pyramid level, VALU / MFMA / LDS busy fractions
at level 1, and the FETCH_SIZE / WRITE_SIZE calibration on a float4 copy of known size (MI355X_MICROARCH.md: on gfx950
FETCH_SIZE reports half the bytes of 16-byte-per-lane reads; every wide load of these kernels is 16 B per lane).
bench.py reads the file for `roofline.traffic`.
    python tools/pmc_traffic.py <pmc dir> [out.json]"""
import csv, glob, json, os, sys
from collections import defaultdict

d = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(root, "profiles", "pmc_traffic.json")


def rows(sub):
    for f in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        yield from csv.DictReader(open(f))


# ---- calibration on known byte counts (tools/microbench.hip): the 2 GiB float4 copy, and the core's own tile pattern
# (16 bytes per lane in 64-byte runs one plane apart: 64 planes x 1088 x 1920 floats = 534,773,760 B each way)
def factor(sub, ctr, match, true_bytes):
    v = [float(r["Counter_Value"]) for r in rows(sub) if r["Kernel_Name"].startswith(match) and r["Counter_Name"] == ctr]
    return true_bytes / (1024.0 * sum(v) / len(v)) if v else None          # true bytes / reported bytes (KB x 1024)
PLANES = 64 * 1088 * 1920 * 4
cal = {"float4_copy": {"fetch": factor("cal_fetch", "FETCH_SIZE", "copy_kernel(", 2 << 30),
                       "write": factor("cal_write", "WRITE_SIZE", "copy_kernel(", 2 << 30)},
       "run64_tile_copy": {"fetch": factor("cal_fetch", "FETCH_SIZE", "run64_copy_kernel", PLANES),
                           "write": factor("cal_write", "WRITE_SIZE", "run64_copy_kernel", PLANES)}}
fetch_k = cal["run64_tile_copy"]["fetch"] or cal["float4_copy"]["fetch"] or 2.0
write_k = cal["run64_tile_copy"]["write"] or cal["float4_copy"]["write"] or 1.0

# ---- the core's launches.  tools/bench_core.py runs level 1, then 2, then 3, the same number of calls each: the k-th
# third of a kernel class's dispatches (in dispatch order) belongs to level k (grid sizes can coincide between levels).
disp = defaultdict(lambda: defaultdict(list))         # (pass, kernel class) -> counter -> [(dispatch id, value)]
for sub in ("fetch", "write", "sq1", "sq2"):
    for r in rows(sub):
        k = r["Kernel_Name"]
        cls = "reduce" if "ss2d_core_kernel<16, 16, 1" in k else "scan" if "ss2d_core_kernel<16, 16, 3" in k else \
              "carry" if "selscan_carry_kernel" in k else "prep" if "ss2d_core_prep_kernel" in k else None
        if cls:
            disp[cls][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
acc = defaultdict(lambda: defaultdict(list))          # (kernel class, level) -> counter -> values
for cls, ctrs in disp.items():
    for ctr, lst in ctrs.items():
        lst.sort()
        n = len(lst) // 3
        for lvl in (1, 2, 3):
            acc[(cls, lvl)][ctr] = [v for _, v in lst[(lvl - 1) * n:lvl * n]]
mean = lambda v: sum(v) / len(v) if v else 0.0
levels = {}
for lvl in (1, 2, 3):
    H, W = 2176 >> lvl, 3840 >> lvl
    e = {"H": H, "W": W, "positions": H * W}
    tot = 0.0
    for cls in ("reduce", "scan", "carry", "prep"):
        c = acc.get((cls, lvl), {})
        fb, wb = mean(c.get("FETCH_SIZE", [])) * 1024 * fetch_k, mean(c.get("WRITE_SIZE", [])) * 1024 * write_k
        e[cls] = {"fetch_bytes": fb, "write_bytes": wb}
        tot += fb + wb
    e["hbm_bytes_per_call"] = tot
    e["bytes_per_position"] = tot / (H * W)
    e["over_3584B"] = tot / (3584.0 * H * W)
    e["over_512B"] = tot / (512.0 * H * W)
    for cls in ("reduce", "scan"):
        c = acc.get((cls, lvl), {})
        gui = mean(c.get("GRBM_GUI_ACTIVE", [])) / 8            # summed over the 8 XCDs
        if gui:
            e[cls]["valu_busy_frac"] = mean(c.get("SQ_ACTIVE_INST_VALU", [])) * 4 / 1024 / gui   # quad-cycles, 1024 SIMDs
            e[cls]["mfma_busy_frac"] = mean(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [])) / 1024 / gui
            e[cls]["lds_busy_frac"] = mean(c.get("SQ_LDS_IDX_ACTIVE", [])) / 256 / gui
    levels[str(lvl)] = e
sys.path.insert(0, root)
import wave_mamba_amd as wm                         # the library the counters were collected on (same tree, same build)
kernels = sorted({r["Kernel_Name"].split("(")[0] for sub in ("fetch", "write") for r in rows(sub)
                  if "ss2d_core" in r["Kernel_Name"] or "selscan_carry" in r["Kernel_Name"]})
res = {
    "build_id": wm._lib.build_id(),
    "kernels": kernels,
    "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/pmc_core.sh); counter value x 1024 x the "
              "calibration factor below (true / reported bytes of copies of known size in tools/microbench, same session)",
    "calibration": {"true_over_reported": cal, "applied": {"fetch": fetch_k, "write": write_k},
                    "note": "applied = the factors of the 64-byte-run tile copy (the core's access pattern)"},
    "ss2d_core": {"levels": levels},
}
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res["calibration"]))
for l, e in levels.items():
    print(f"level {l}: {e['hbm_bytes_per_call'] / 1e9:.3f} GB per call = {e['bytes_per_position']:.0f} B/position "
          f"({e['over_3584B']:.2f} x 3584 B, {e['over_512B']:.2f} x 512 B)",
          {k: round(v, 3) for k, v in e.get("scan", {}).items() if k.endswith("frac")})
