#!/usr/bin/env python3
"""Which kernels wait out their global loads one at a time?  Compiles the library with -save-temps and, per kernel, counts
global / buffer loads and `s_waitcnt vmcnt(0)` instructions: a ratio near 1 means load - full wait - load - full wait
(the `cond ? load : 0` pattern: a branch around each load and a wait behind it).

usage: python tools/isa_load_waits.py [min-loads]"""
import re, sys, subprocess, os, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp()
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value",
                "-save-temps=obj", os.path.join(root, "wave_mamba_amd/csrc/wavemamba_hip.hip"), "-o", os.path.join(tmp, "t.so")],
               cwd=tmp, capture_output=True)
asm = [f for f in os.listdir(tmp) if f.endswith(".s") and "gfx950" in f][0]
cur, stats = None, {}
for ln in open(os.path.join(tmp, asm)):
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        cur = m.group(1); stats[cur] = [0, 0]; continue
    if cur is None: continue
    t = ln.strip()
    if t.startswith("global_load") or t.startswith("buffer_load"): stats[cur][0] += 1
    elif t.startswith("s_waitcnt") and "vmcnt(0)" in t: stats[cur][1] += 1
    elif t.startswith("s_endpgm"): cur = None
names = list(stats)
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
minl = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rows = [(stats[n][1] / max(1, stats[n][0]), stats[n], d) for n, d in zip(names, dem) if stats[n][0] >= minl]
for r, st, d in sorted(rows, reverse=True):
    print(f"{r:5.2f}  loads {st[0]:4d}  full waits {st[1]:4d}  {d[:130]}")
