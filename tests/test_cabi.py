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


def test_core_work_split_is_consistent():
    """wm_ss2d_core_plan (host-only): the work split wm_ss2d_core_fwd uses covers the map exactly once in every direction,
    is the same on every call (the search result is cached per shape), and the UHD level-1 launch is not the 480 equal
    workgroups of round 2 (two dispatch rounds, the second 7/8 full) but column workgroups + shorter row workgroups."""
    import ctypes
    from wave_mamba_amd import _lib
    lib = _lib.load()
    out = (ctypes.c_int * 10)()
    for (B, H, W, N) in [(1, 1088, 1920, 16), (1, 544, 960, 16), (1, 272, 480, 16), (1, 2048, 2048, 32), (8, 256, 256, 16),
                         (2, 24, 40, 16), (1, 8, 12, 16), (1, 33, 72, 32), (1, 16, 2048, 16), (1, 8192, 16, 16),
                         (1, 10, 14, 16), (1, 9, 7, 32), (2, 33, 71, 16), (1, 1, 1, 16)]:      # odd widths: same plan, element-wise tiles
        assert lib.wm_ss2d_core_plan(B, 64, H, W, N, 2, out) == 0
        nw, seg, nseg, ctiles, cwgs, rchunk, rnchunks, rwgs, grid, span = list(out)
        L = H * W
        assert nw == (16 if N <= 16 else 8)
        assert seg % 16 == 0 and seg * nseg >= H and seg * (nseg - 1) < H            # column segments tile the rows
        assert ctiles * nw >= W and (ctiles - 1) * nw < W and cwgs >= ctiles * nseg and cwgs % 8 == 0
        assert rchunk % 16 == 0 and rchunk * rnchunks >= L and rchunk * (rnchunks - 1) < L
        assert rwgs * nw >= rnchunks and grid == B * (2 * rwgs + 2 * cwgs) and span > 0
        again = (ctypes.c_int * 10)()
        assert lib.wm_ss2d_core_plan(B, 64, H, W, N, 2, again) == 0 and list(again) == list(out)
    assert lib.wm_ss2d_core_plan(1, 64, 1088, 1920, 16, 2, out) == 0
    assert out[2] == 1 and out[5] // 16 < 1088 // 16            # one segment per column; row chunks shorter than a column
    assert lib.wm_ss2d_core_plan(1, 64, 8, 8, 64, 2, out) == -5        # N > 32
    assert lib.wm_ss2d_core_prep_bytes(16) > 0 and lib.wm_ss2d_core_prep_bytes(33) == 0


def test_core_plan_cache_keeps_serving_new_shapes():
    """the plan cache stopped inserting at 256 shapes, so every later shape re-ran the list-scheduling search on
    every call (5-18 ms under a global mutex, 14 core calls per image).  It is an LRU now: after far more distinct shapes
    than it holds, a NEW shape is searched once and then served from the cache, and plans are unchanged by eviction."""
    import ctypes
    import time
    from wave_mamba_amd import _lib
    lib = _lib.load()
    out, again = (ctypes.c_int * 10)(), (ctypes.c_int * 10)()
    assert lib.wm_ss2d_core_plan(1, 64, 272, 480, 16, 2, out) == 0
    first = list(out)
    for i in range(1300):                                        # > the cache's 1024 entries, all distinct (small maps: fast searches)
        assert lib.wm_ss2d_core_plan(1, 64, 16 + 8 * (i % 50), 16 + 4 * (i // 50), 16, 2, again) == 0
    t0 = time.perf_counter()
    assert lib.wm_ss2d_core_plan(1, 64, 1096, 1928, 16, 2, again) == 0          # a new, UHD-sized shape: searched now
    cold = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(20):
        assert lib.wm_ss2d_core_plan(1, 64, 1096, 1928, 16, 2, out) == 0 and list(out) == list(again)
    warm = (time.perf_counter() - t0) / 20
    assert warm < 2e-3 and warm < 0.5 * max(cold, 4e-3), (cold, warm)           # cached: no search per call
    assert lib.wm_ss2d_core_plan(1, 64, 272, 480, 16, 2, out) == 0 and list(out) == first   # evicted and re-planned: same plan


def test_library_isa_has_no_scalar_source_packed_f32_with_routed_halves():
    """tools/lint_packed_f32.py on the built library: `v_pk_{fma,mul,add}_f32` with a scalar source (SGPR pair / inline constant)
    AND a VGPR source read through op_sel = 1 returns a zero half in lanes 48..63 on MI355X while LDS-fed MFMAs of another kernel
    share the SIMD (tools/ubench_pk_coexec.hip, profiles/r05/) - the multi-stream mismatch of rounds 4-5.  The library must not
    contain the form; the lint's pattern matcher is checked on the instruction that did the damage."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import lint_packed_f32 as lint
    finally:
        sys.path.pop(0)
    sample = ("0000000000001000 <_ZN2wm16dwconv3x3_kernelILi1ELb1EtEEvPKT1_PKfS5_PS1_iiix>:\n"
              "\tv_pk_fma_f32 v[42:43], s[48:49], v[20:21], v[42:43] op_sel:[0,0,1] op_sel_hi:[1,1,0] // 000000001000: D3B0402A 1CAA2830\n"
              "\tv_pk_fma_f32 v[42:43], s[76:77], v[24:25], v[42:43]\n"
              "\tv_pk_fma_f32 v[50:51], v[26:27], v[48:49], v[50:51] op_sel:[0,0,1] op_sel_hi:[1,1,0]\n"
              "\tv_pk_mul_f32 v[48:49], v[48:49], s[18:19] op_sel_hi:[1,0]\n"
              "\tv_pk_fma_f32 v[6:7], v[8:9], 2.0, v[6:7] op_sel:[0,0,1] op_sel_hi:[1,0,0]\n"
              "\tv_pk_add_f32 v[10:11], v[12:13], v[10:11] op_sel:[0,1] op_sel_hi:[1,0]\n"
              "\tv_pk_mul_f32 v[14:15], v[14:15], v[16:17] op_sel:[1,0]\n"
              "\tv_pk_fma_f32 v[18:19], s[2:3], v[20:21], v[18:19] op_sel_hi:[1,0,1]\n")
    hits = lint.offending(sample)
    # flagged: the dwconv3x3<bf16> form, (conservatively) the all-VGPR swap, the inline-constant form, round 4's v_pk_add_f32
    assert [h[1].split()[1] for h in hits] == ["v[42:43],", "v[50:51],", "v[6:7],", "v[10:11],"], hits
    if not os.path.exists(os.path.join(lint.LLVM, "llvm-objdump")):
        pytest.skip("llvm-objdump not available")
    bad = lint.offending(lint.disassemble(_lib.LIB_PATH))
    assert not bad, f"{len(bad)} packed-fp32 instruction(s) with unsafe op_sel routing, first: {bad[0]}"


def test_zero_arena_registry_bookkeeping():
    """wm_zero_arena_register / _unregister (include/wavemamba_hip.h): pure host bookkeeping, so it runs without a GPU - empty and
    overlapping ranges and unknown bases are refused, a range can be registered again after it was unregistered."""
    lib = _lib.load()
    base = 0x7f0000000000
    assert lib.wm_zero_arena_register(None, 4096) == _lib.WM_ENULL
    assert lib.wm_zero_arena_register(base, 0) == _lib.WM_EINVAL
    assert lib.wm_zero_arena_register(base, 4096) == _lib.WM_OK
    try:
        assert lib.wm_zero_arena_register(base + 4000, 4096) == _lib.WM_EINVAL          # overlaps the tail
        assert lib.wm_zero_arena_register(base - 100, 200) == _lib.WM_EINVAL            # overlaps the head
        assert lib.wm_zero_arena_register(base + 4096, 4096) == _lib.WM_OK              # adjacent: fine
        assert lib.wm_zero_arena_unregister(base + 4096) == _lib.WM_OK
        assert lib.wm_zero_arena_unregister(base + 8) == _lib.WM_EINVAL                 # not a base
    finally:
        assert lib.wm_zero_arena_unregister(base) == _lib.WM_OK
    assert lib.wm_zero_arena_unregister(base) == _lib.WM_EINVAL
    assert lib.wm_zero_arena_register(base, 64) == _lib.WM_OK and lib.wm_zero_arena_unregister(base) == _lib.WM_OK
