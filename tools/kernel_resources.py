#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS usage of the library (hipcc -Rpass-analysis=kernel-resource-usage), one line each.

usage: python tools/kernel_resources.py [substring-of-demangled-name] [extra hipcc flags ...]
"""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flt = sys.argv[1] if len(sys.argv) > 1 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value",
       "-Rpass-analysis=kernel-resource-usage", os.path.join(root, "wave_mamba_amd/csrc/wavemamba_hip.hip"),
       "-o", "/tmp/_kres.so"] + sys.argv[2:]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r"remark: (?:\S+: )?\s*Function Name: (\S+)", line) or re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/\w+\])?: (\d+)", line)
    if m and cur: rows[cur][m.group(1).strip()] = int(m.group(2))
names = list(rows)
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
for n, d in zip(names, dem):
    if flt and flt not in d: continue
    r = rows[n]
    print(f"{d[:110]:110s} vgpr {r.get('VGPRs', -1):4d} agpr {r.get('AGPRs', -1):3d} sspill {r.get('SGPRs Spill', -1):3d} "
          f"vspill {r.get('VGPRs Spill', -1):3d} scratch {r.get('ScratchSize', -1):4d} occ {r.get('Occupancy', -1):2d} lds {r.get('LDS Size', -1)}")
