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


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libwavemamba_hip.so")


def source_id():
    """sha256 over the library's sources (csrc/*.hip, csrc/*.hip.h, include/wavemamba_hip.h), first 16 hex digits: compiled
    into the library (wm_build_id) so that measurements taken on one binary (profiles/pmc_traffic.json) are never quoted
    for another."""
    import hashlib
    h = hashlib.sha256()
    for d in DEPS:
        with open(d, "rb") as f:
            h.update(os.path.basename(d).encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=True):
    """Compile csrc/wavemamba_hip.hip for gfx950 -> libwavemamba_hip.so.  Returns the path."""
    if not force and not is_stale():
        return LIB
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wno-unused-value", f'-DWM_BUILD_ID="{source_id()}"', SRC, "-o", LIB + ".tmp"]
    if verbose:
        print("[wave_mamba_amd] " + " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
