"""CPU: the C-ABI shared library loads and exports every symbol include/wavemamba_hip.h declares.
No compute calls (no GPU here) - only host-side entry points (version, strerror, workspace sizing,
argument validation that returns before any launch)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
import wave_mamba_amd as wm
from wave_mamba_amd import _lib


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "wavemamba_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(wm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f"libwavemamba_hip.so does not export {n}"
    assert set(_lib.SIGNATURES) == set(names), "ctypes binding and header disagree"


def test_host_side_entry_points():
    lib = _lib.load()
    assert lib.wm_abi_version() == _lib.ABI_VERSION
    assert lib.wm_strerror(0) == b"ok"
    assert b"shape" in lib.wm_strerror(-1)
    # workspace sizing is pure host arithmetic: UHD level-1 scan (B=1, KD=256, L=2,088,960)
    ws = lib.wm_selscan_fwd_workspace_bytes(1, 256, 2088960, 16, 4)
    assert 0 < ws < 512 * 2 ** 20
    assert lib.wm_selscan_fwd_workspace_bytes(1, 256, 32, 16, 4) == 0       # single chunk
    assert lib.wm_selscan_fwd_workspace_bytes(1, 256, 1024, 64, 4) == 0      # N > 32 unsupported -> 0


def test_argument_validation_returns_before_launch():
    lib = _lib.load()
    # odd H: the reference raises RuntimeError (strided slices disagree); here WM_EINVAL
    assert lib.wm_dwt2d_fwd(None, None, None, None, None, 1, 1, 5, 4, 0, None) == -1
    # NULL tensors
    assert lib.wm_dwt2d_fwd(None, None, None, None, None, 1, 1, 4, 4, 0, None) == -2
    assert lib.wm_selscan_fwd(*([None] * 10), None, 0, 1, 8, 16, 4, 2, 1, None) == -2
    assert lib.wm_selscan_fwd(*([None] * 10), None, 0, 1, 8, 16, 4, 3, 1, None) == -1   # dim % G
    assert lib.wm_selscan_fwd(*([None] * 10), None, 0, 1, 8, 16, 40, 2, 1, None) == -5  # N > 32
    # the 3x3 kernel selector is host state: 0 / 1 / 2 accepted, anything else WM_EINVAL (and the mode stays)
    assert lib.wm_conv2d_select(3) == -1 and lib.wm_conv2d_select(-1) == -1
    assert lib.wm_conv2d_select(2) == 0 and lib.wm_conv2d_select(1) == 0 and lib.wm_conv2d_select(0) == 0
    # convolution arguments are checked before any device work: kernel size, NULLs, fragment alignment
    assert lib.wm_conv2d_fwd(*([None] * 8), 1, 32, 0, 0, 32, 8, 8, 5, None) == -5           # ks = 5
    assert lib.wm_conv2d_fwd(*([None] * 8), 1, 32, 0, 0, 32, 8, 8, 3, None) == -2           # NULL tensors
    assert lib.wm_conv2d_fwd(*([None] * 8), 0, 32, 0, 0, 32, 8, 8, 3, None) == 0            # empty batch
    # empty problems are a no-op
    assert lib.wm_dwt2d_fwd(None, None, None, None, None, 0, 3, 4, 4, 0, None) == 0
    assert lib.wm_selscan_fwd(*([None] * 10), None, 0, 0, 8, 16, 4, 2, 1, None) == 0


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libwavemamba_hip.so")
    with pytest.raises(_lib.WaveMambaHipError):
        _lib.load()
