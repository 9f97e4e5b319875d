#!/usr/bin/env python3
"""ISA lint of libwavemamba_hip.so: packed-fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) whose op_sel routing
is not safe on MI355X.

What was measured (round 5; standalone reproducer tools/ubench_pk_coexec.hip + tools/repro_pk_micro.py, 1.05e10 eight-step chains
per form, results checked against integer arithmetic; profiles/r05/ubench_pk_coexec.txt): while another wave on the SIMD runs
LDS-fed MFMAs (`ds_read_b128` + `v_mfma_f32_32x32x16_bf16` on random data - any 3x3 matrix-core convolution on a second stream)
the following forms return a ZERO for the routed half in lanes 48..63 (the wave's last 16-lane pass), 1e-8 .. 1e-3 of the time:
    v_pk_fma_f32 v, s[..], v, v   op_sel:[0,0,1] op_sel_hi:[1,1,0]      scalar source + SWAPPED VGPR source   (dwconv3x3<bf16>, SLP)
    v_pk_fma_f32 v, v, 2.0, v     op_sel:[0,0,1] op_sel_hi:[1,0,0]      inline constant + swapped VGPR source (haar_analysis, SLP)
    v_pk_add_f32 v, v, v          op_sel:[0,1]   op_sel_hi:[1,0]        VGPR sources only, one SWAPPED         (round 4's core backward)
and these never do (0 of 1.05e10 each, same aggressors):
    the same v_pk_fma_f32 with three VGPR sources; any form without op_sel; (low, low) broadcasts (op_sel_hi = 0 alone) with or
    without a scalar source; v_pk_mul_f32 v, v, v op_sel:[1,0] ((high, high) broadcast - the scans' form).
Alone on the chip every form is exact.  The multi-stream mismatch of rounds 4-5 was the first form (compiler-generated, in
dwconv3x3<bf16>, under the side streams' 3x3 convolutions); round 4's "v_pk_add_f32 op_sel fault" in the core backward was the third.

Rule enforced here (conservative): no packed-fp32 instruction may
    (A) read a source SWAPPED (op_sel = 1 and op_sel_hi = 0 for that source), whatever the other sources are, or
    (B) combine a scalar source (SGPR pair / inline constant / literal) with a VGPR source read through op_sel = 1.
The library is built with -fno-slp-vectorize (the vectoriser produced two of the three); wave_mamba_amd/build.py runs this lint
at the end of every build and tests/test_cabi.py runs it on the library the tests load.

usage: python tools/lint_packed_f32.py [path/to/lib.so]     exit status 1 when such an instruction exists
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
    """-> [(symbol, instruction)] breaking rule (A) or (B) of the module docstring"""
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
        for i, s in enumerate(srcs[:len(sel)]):
            swapped = sel[i] == 1 and selhi[i] == 0                                   # (A)
            routed_next_to_scalar = any(scalar) and not scalar[i] and sel[i] == 1     # (B)
            if swapped or routed_next_to_scalar:
                out.append((sym, line.split("//")[0].strip()))
                break
    return out


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "libwavemamba_hip.so")
    bad = offending(disassemble(lib))
    for sym, ins in bad[:40]:
        print(f"{sym}: {ins}")
    print(f"{len(bad)} packed-fp32 instruction(s) with unsafe op_sel routing in {lib}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
