#!/usr/bin/env python3
"""Per-kernel HBM traffic and busy fractions of ONE UHD forward from the five rocprofv3 --pmc passes of tools/pmc_step_kernel.sh run
over every kernel of the step (kernel regex 'wm::'):  python tools/pmc_step_table.py <dir> [forwards-per-run]
FETCH_SIZE / WRITE_SIZE are in KiB on this stack; the factors true / reported (2.00 / 1.00, float4 streaming copy of known size in
tools/microbench, MI355X_MICROARCH.md's gfx950 correction) are applied; the scan core's 64-byte-run pattern has its own factors in
profiles/pmc_traffic.json and is quoted from there."""
import csv, glob, os, re, sys
from collections import defaultdict
d = sys.argv[1]
fw = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0          # forwards per profiled run; 0 = from the Haar analysis launches (3 per forward)


def short(k):
    k = k.split("(")[0].replace("void ", "").replace("wm::", "")
    return re.sub(r"\s+", "", k)[:64]


tot = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = short(row["Kernel_Name"]); c = row["Counter_Name"]
        tot[k][c] += float(row["Counter_Value"]); cnt[k][c] += 1
if fw <= 0:
    ha = [max(c.values()) for k, c in cnt.items() if k.startswith("haar_analysis")]
    fw = sum(ha) / 3.0 if ha else 3.0
POS = 7311360          # scanned positions per UHD forward (14 LFSSBlocks), bench.py
ALGO = {               # algorithmic GB per forward where bench.py defines them (SURVEY 8d / bench.py LFSS_BYTES_PER_POS)
    "lfss_in_mfma": 384 * POS / 1e9, "lfss_mid_mfma": 1536 * POS / 1e9, "lfss_out_conv_mfma": 512 * POS / 1e9,
    "dwconv3x3_kernel<1": 512 * POS / 1e9, "haar_analysis": 2.80756224, "haar_synthesis": 2.80756224,
}
rows = []
for k, cs in tot.items():
    n = max(cnt[k].values()) / fw
    fetch = cs.get("FETCH_SIZE", 0.0) * 1024 * 2.0 / fw / 1e9
    write = cs.get("WRITE_SIZE", 0.0) * 1024 * 1.0 / fw / 1e9
    wc = cs.get("SQ_WAVE_CYCLES", 0.0); gui = cs.get("GRBM_GUI_ACTIVE", 0.0)
    fr = lambda name: (cs.get(name, 0.0) / wc) if wc else float("nan")
    # chip-wide busy fractions as tools/pmc_traffic.py forms them: quad-cycles over 1024 SIMDs x GUI-active cycles (the GUI counter
    # comes back once per XCD: / 8; cross-check: the scan core's VALU busy is 0.73-0.75 in profiles/pmc_traffic.json)
    valu = cs.get("SQ_ACTIVE_INST_VALU", 0.0) * 4 / 1024 / (gui / 8) if gui else float("nan")
    mfma = cs.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024 / (gui / 8) if gui else float("nan")
    rows.append((fetch + write, k, n, fetch, write, valu, fr("SQ_WAIT_ANY"), fr("SQ_ACTIVE_INST_VMEM"), mfma))
rows.sort(reverse=True)
print(f"# per UHD forward (1x3x2176x3840, fp32 planes): launches, HBM GB fetched / written (PMC, corrected), VALU / MFMA busy chip-wide (of GRBM_GUI_ACTIVE), waiting / VMEM per wave (of SQ_WAVE_CYCLES)")
print(f"# forwards in the profiled run: {fw:g}")
print(f"{'kernel':66s} {'calls':>6s} {'fetch GB':>9s} {'write GB':>9s} {'algo GB':>8s} {'x algo':>7s} {'VALU busy':>9s} {'waiting':>8s} {'MFMA busy':>9s}")
for t, k, n, fe, wr, va, wa, vm, mf in rows:
    al = next((v for a, v in ALGO.items() if k.startswith(a)), None)
    print(f"{k:66s} {n:6.1f} {fe:9.3f} {wr:9.3f} " + (f"{al:8.3f} {(fe + wr) / al:7.2f}" if al else f"{'':8s} {'':7s}")
          + f" {va:9.2f} {wa:8.2f} {mf:9.3f}")
print(f"{'sum':66s} {'':6s} {sum(r[3] for r in rows):9.3f} {sum(r[4] for r in rows):9.3f}")
