#!/usr/bin/env python3
"""ISA lint of libwavemamba_hip.so: no packed-fp32 VALU instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) may combine a
SCALAR source (SGPR pair / inline constant) with a VGPR source read through op_sel = 1 (low result lane <- HIGH half).

Why (round 5, tools/ubench_pk_coexec.hip + tools/repro_pk_micro.py, profiles/r05/): on MI355X
    v_pk_fma_f32 v[d:d+1], s[a:a+1], v[b:b+1], v[c:c+1] op_sel:[0,0,1] op_sel_hi:[1,1,0]
returns a ZERO for the re-routed half in lanes 48..63, now and then, while another wave on the SIMD runs LDS-fed MFMAs (a 3x3
convolution on another stream).  Measured boundaries of the form (1e10 eight-step chains each, profiles/r05/ubench_pk_coexec.txt):
an SGPR pair OR an inline constant as the scalar source - wrong; the same routing with every source in VGPRs - never; a scalar
source with a (low, low) broadcast of a VGPR source (op_sel_hi = 0 alone) - never; no scalar source, no routing - never.
The compiler's SLP vectoriser produced the form in dwconv3x3<bf16> (the multi-stream mismatch of rounds 4-5) and once, with a
constant, in haar_analysis; the library is built with -fno-slp-vectorize since and this lint runs at the end of every build.

usage: python tools/lint_packed_f32.py [path/to/lib.so]     exit status 1 when an instruction of that form exists
"""
import os, re, subprocess, sys, tempfile, glob, shutil

LLVM = "/opt/rocm/lib/llvm/bin"
PK = re.compile(r"^\s*(v_pk_(?:fma|mul|add)_f32)\s+(.*?)(?:\s*//.*)?$")
SEL = re.compile(r"op_sel:\[([01,]+)\]")
SELHI = re.compile(r"op_sel_hi:\[([01,]+)\]")


def disassemble(lib):
    tmp = tempfile.mkdtemp(prefix="wm_lint_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, capture_output=True, cwd=tmp)
        objs = [p for p in glob.glob(local + ".*") if "amdgcn" in p]
        if not objs:
            raise RuntimeError("no device code object in " + lib)
        return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", objs[0]], check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def offending(text):
    """-> [(symbol, instruction)]: packed-fp32 ops with a scalar source AND a VGPR source read through op_sel = 1"""
    out, sym = [], "?"
    for line in text.splitlines():
        if line.endswith(">:"):
            sym = line.split("<")[-1][:-2]
            continue
        m = PK.match(line)
        if not m:
            continue
        ops = [o.strip() for o in m.group(2).split(" op_sel")[0].split(",")]
        srcs = ops[1:]                                       # (vdst first)
        n = len(srcs)
        sel = [int(v) for v in SEL.search(line).group(1).split(",")] if SEL.search(line) else [0] * n
        selhi = [int(v) for v in SELHI.search(line).group(1).split(",")] if SELHI.search(line) else [1] * n
        scalar = [not s.startswith(("v[", "v", "a[")) for s in srcs]
        if not any(scalar):
            continue
        for i, s in enumerate(srcs):
            if not scalar[i] and sel[i] != 0:
                out.append((sym, line.split("//")[0].strip()))
                break
    return out


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                            "wave_mamba_amd", "libwavemamba_hip.so")
    bad = offending(disassemble(lib))
    for sym, ins in bad[:40]:
        print(f"{sym}: {ins}")
    print(f"{len(bad)} packed-fp32 instruction(s) with a scalar source and an op_sel-routed VGPR source in {lib}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
