#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the rocprofv3 --pmc passes of tools/pmc_core.sh (fused SS2D core, UHD level 1):
HBM bytes per launch (FETCH_SIZE + WRITE_SIZE, KB x 1024) against the algorithmic bytes of DESIGN.md section 4, VALU /
MFMA busy fractions.  bench.py reads the file for `roofline.traffic`.
    python tools/pmc_traffic.py <pmc dir> [out.json]"""
import csv, glob, json, os, sys
from collections import defaultdict
d = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
L, D = 1088 * 1920, 64
classes = {   # bench class -> (kernel substrings, algorithmic bytes per launch)
    "ss2d_proj": (["ss2d_proj_kernel"], L * (256 + 4 * 144)),
    "ss2d_row_reduce": (["ss2d_row_kernel<1"], L * (256 + 80)),
    "ss2d_row_scan": (["ss2d_row_kernel<3"], L * (256 + 144 + 256)),
    "ss2d_col_reduce": (["ss2d_col_kernel<1"], L * (256 + 80)),
    "ss2d_col_scan": (["ss2d_col_kernel<3"], L * (256 + 144 + 256)),
}
res = {}
for cls, (subs, algo) in classes.items():
    ks = [k for k in acc if any(s in k for s in subs)]
    if not ks:
        continue
    mean = lambda c: sum(sum(acc[k][c]) / len(acc[k][c]) for k in ks if acc[k].get(c)) / max(1, sum(1 for k in ks if acc[k].get(c)))
    fetch, write = mean("FETCH_SIZE") * 1024, mean("WRITE_SIZE") * 1024
    gui = mean("GRBM_GUI_ACTIVE") / 8                      # summed over the 8 XCDs
    valu = mean("SQ_ACTIVE_INST_VALU") * 4 / 1024 / gui if gui else None    # quad-cycles, 1024 SIMDs
    mfma = mean("SQ_VALU_MFMA_BUSY_CYCLES") / 1024 / gui if gui else None
    res[cls] = {"shape": "UHD level 1 (B=1, D=64, 1088x1920), one launch (mean of the two directions)",
                "pmc_fetch_bytes": fetch, "pmc_write_bytes": write, "hbm_bytes_per_launch": fetch + write,
                "algorithmic_bytes_per_launch": algo, "traffic_over_algorithmic": (fetch + write) / algo,
                "valu_busy_frac": valu, "mfma_busy_frac": mfma,
                "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KB x1024), separate passes (tools/pmc_core.sh); FETCH_SIZE can "
                        "under-count wide streaming reads on gfx950 by up to 2x (MI355X_MICROARCH.md): uncalibrated"}
json.dump(res, open(out, "w"), indent=1)
for k, v in res.items():
    print(f"{k:16s} traffic/algorithmic {v['traffic_over_algorithmic']:.3f}  valu {v['valu_busy_frac']:.3f}  mfma {v['mfma_busy_frac']:.3f}")
