"""Build the gfx950 HIP library (and nothing else) in-tree with hipcc.

    python -m wave_mamba_amd.build         # or: python wave_mamba_amd/build.py

hipcc cross-compiles for gfx950 without a GPU; the resulting libwavemamba_hip.so sits next to this
file (git-ignored, but it travels to the GPU box with the gpurun snapshot).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "wavemamba_hip.hip")
DEPS = [SRC] + sorted(os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc"))
                      if f.endswith(".hip.h")) + [os.path.join(HERE, "..", "include", "wavemamba_hip.h")]
LIB = os.path.join(HERE, "libwavemamba_hip.so")


# -fno-slp-vectorize: the SLP vectoriser pairs independent scalar fp32 operations into packed `v_pk_*_f32` instructions and
# routes halves with op_sel as it likes.  One form it produced - a packed-fp32 op with a SCALAR source and a VGPR source read
# through op_sel = 1 (`v_pk_fma_f32 v, s[..], v, v op_sel:[0,0,1] op_sel_hi:[1,1,0]` in dwconv3x3<bf16>) - returns a zero for the
# re-routed half in lanes 48..63 on MI355X while LDS-fed MFMAs of another kernel share the SIMD: the multi-stream mismatch of
# rounds 4-5 (standalone reproducer tools/ubench_pk_coexec.hip, evidence profiles/r05/).  The packed arithmetic this library
# wants (the scans' state pairs) is written as two-element vectors in the source and is not the vectoriser's work.
# tools/lint_packed_f32.py disassembles the result and refuses a library that contains the form, whoever produced it.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value", "-fno-slp-vectorize"]


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libwavemamba_hip.so")


def source_id():
    """sha256 over the compiler flags and the library's sources (csrc/*.hip, csrc/*.hip.h, include/wavemamba_hip.h), first 16 hex digits: compiled
    into the library (wm_build_id) so that measurements taken on one binary (profiles/pmc_traffic.json) are never quoted
    for another."""
    import hashlib
    h = hashlib.sha256()
    h.update(" ".join(HIPCC_FLAGS).encode())             # (a different code generator is a different binary)
    for d in DEPS:
        with open(d, "rb") as f:
            h.update(os.path.basename(d).encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in DEPS)


def lint(lib=LIB):
    """_lint_packed_f32.py (same directory) on the built library; raises when the vulnerable instruction form is present.  (Skipped, with a
    note, where the LLVM binutils are absent - the driver image has them.)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("wave_mamba_amd._lint_packed_f32", os.path.join(HERE, "_lint_packed_f32.py"))
    lint_packed_f32 = importlib.util.module_from_spec(spec)        # (by path: this file also runs as a plain script)
    spec.loader.exec_module(lint_packed_f32)
    if not os.path.exists(os.path.join(lint_packed_f32.LLVM, "llvm-objdump")):
        print("[wave_mamba_amd] llvm-objdump not found: ISA lint skipped", file=sys.stderr)
        return None
    bad = lint_packed_f32.offending(lint_packed_f32.disassemble(lib))
    if bad:
        raise RuntimeError(f"{lib}: {len(bad)} packed-fp32 instruction(s) with unsafe op_sel routing "
                           f"(first: {bad[0][0]}: {bad[0][1]}) - see tools/lint_packed_f32.py")
    return 0


def build(force=False, verbose=True):
    """Compile csrc/wavemamba_hip.hip for gfx950 -> libwavemamba_hip.so (+ the ISA lint).  Returns the path."""
    if not force and not is_stale():
        return LIB
    cmd = [hipcc_path()] + HIPCC_FLAGS + [f'-DWM_BUILD_ID="{source_id()}"', SRC, "-o", LIB + ".tmp"]
    if verbose:
        print("[wave_mamba_amd] " + " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    lint(LIB + ".tmp")
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
