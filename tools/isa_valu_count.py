#!/usr/bin/env python3
"""VALU issue cycles of the innermost (scan-step) loops of the selective-scan kernels, counted from the ISA (VERDICT r4 item 2:
"count the issue cycles from the ISA, do not estimate").

usage: python tools/isa_valu_count.py [file.s]      (default: compiles csrc/wavemamba_hip.hip -S with the library's flags)

Per kernel instantiation and per innermost loop (the `for q` loop of core_body: FOUR scan steps per iteration): instructions by
class and the issue cycles they cost one wave on its SIMD (gfx950, tools/microbench + ubench_mfma_valu: a VALU instruction of
a wave64 occupies the SIMD's issue port for 4 cycles, a transcendental - v_exp / v_log / v_rcp / v_rsq / v_sqrt - for 8; DS, VMEM,
SALU and MFMA instructions issue on other ports).  floor(ms) of a UHD step = cycles per step x wave-steps / (1024 SIMDs x clock).
"""
import os, re, subprocess, sys, json, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRANS = re.compile(r"^v_(exp|log|rcp|rsq|sqrt|sin|cos)_")
VALU = re.compile(r"^v_")
NOT_VALU = re.compile(r"^v_(mfma|smfmac|accvgpr)")


def isa(path=None):
    if path:
        return open(path).read()
    sys.path.insert(0, ROOT)
    from wave_mamba_amd import build
    flags = [f for f in build.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    out = "/tmp/wm_isa_count.s"
    subprocess.run([build.hipcc_path()] + flags + ["-S", "--cuda-device-only", "-o", out, build.SRC], check=True, capture_output=True)
    return open(out).read()


def kernels(text, pattern):
    cur, buf = None, []
    for line in text.splitlines():
        m = re.match(r"^(_ZN2wm\w+):", line)
        if m:
            cur, buf = m.group(1), []
            continue
        if cur and line.startswith(".Lfunc_end"):
            if re.search(pattern, cur):
                yield cur, buf
            cur = None
            continue
        if cur:
            buf.append(line)


def inner_loops(lines):
    """-> {header label: [instruction mnemonics]} for every innermost loop (blocks annotated `in Loop: Header=X` with the
    largest depth that have no deeper child)."""
    blocks, cur_hdr, cur_depth = collections.defaultdict(list), None, 0
    depth_of, parents = {}, set()
    for line in lines:
        m = re.search(r";\s+(?:in Loop: Header=(\w+) Depth=(\d+)|=>\s*This (?:Inner )?Loop Header: Depth=(\d+))", line)
        lab = re.match(r"^\.?(L?BB\d+_\d+):", line)
        if "Parent Loop" in line:
            pm = re.search(r"Parent Loop (\w+)", line)
            if pm:
                parents.add(pm.group(1))
            continue
        if m:
            if m.group(1):
                cur_hdr, cur_depth = m.group(1), int(m.group(2))
            else:
                cur_hdr, cur_depth = (lab.group(1).lstrip("L") if lab else cur_hdr), int(m.group(3))
            depth_of[cur_hdr] = cur_depth
            continue
        if lab and not m:
            cur_hdr = None                                   # a block outside every loop
            continue
        s = line.strip()
        if cur_hdr and s and not s.startswith((";", ".")):
            blocks[cur_hdr].append(s.split()[0])
    return {h: ins for h, ins in blocks.items() if h not in parents}


def classify(ins):
    c = collections.Counter()
    for i in ins:
        if NOT_VALU.match(i):
            c["mfma"] += 1
        elif TRANS.match(i):
            c["trans"] += 1
        elif i.startswith("v_pk_"):
            c["valu_packed"] += 1
        elif VALU.match(i):
            c["valu_other"] += 1
        elif i.startswith("ds_"):
            c["lds"] += 1
        elif i.startswith(("global_", "buffer_", "flat_")):
            c["vmem"] += 1
        elif i.startswith("s_"):
            c["salu"] += 1
    c["valu_issue_cycles"] = 4 * (c["valu_packed"] + c["valu_other"]) + 8 * c["trans"]
    return dict(c)


def main():
    text = isa(sys.argv[1] if len(sys.argv) > 1 else None)
    out = {}
    for name, lines in kernels(text, r"ss2d_core_kernelILi16ELi16ELi[13]ELb0EfLb1EEE"):
        phase = "reduce" if "ILi16ELi16ELi1E" in name else "scan"
        loops = inner_loops(lines)
        # the step loops are the innermost loops with transcendentals in them; one per direction variant (row / column x forward / reversed)
        steps = [classify(v) for v in loops.values() if any(TRANS.match(i) for i in v) and len(v) > 100]
        out[phase] = {"kernel": name, "step_loops": steps,
                      "valu_issue_cycles_per_step": sum(s["valu_issue_cycles"] for s in steps) / len(steps) / 4.0 if steps else None}
    pos = 7311360                      # scanned positions per UHD image (0.875 H W of the padded frame)
    wave_steps = pos * 4               # four directions, one wave (64 channels) per position and direction
    cyc = sum(out[p]["valu_issue_cycles_per_step"] for p in out)
    for clock in (2.4e9, 2.1e9):
        out[f"valu_floor_ms_at_{clock / 1e9:.1f}GHz"] = cyc * wave_steps / (1024 * clock) * 1e3
    out["note"] = ("both passes of wm_ss2d_core_fwd, innermost step loops only (four steps per iteration, averaged over the four "
                   "direction variants): projection, staging, y stores, prologues and tails are NOT in the floor")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
