"""ctypes binding of libwavemamba_hip.so (C ABI declared in include/wavemamba_hip.h).

The product path has NO fallback: if the library is missing or fails to load, every op raises.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WAVEMAMBA_HIP_LIB") or os.path.join(HERE, "libwavemamba_hip.so")  # env: A/B builds

WM_F32, WM_BF16 = 0, 1
WM_OK, WM_EINVAL, WM_ENULL, WM_EALIGN, WM_EWORKSPACE, WM_EUNSUPPORTED, WM_EHIP = 0, -1, -2, -3, -4, -5, -6
WM_PROF_NKERNELS = 20
ABI_VERSION = 31

_c = ctypes
_p, _i, _i64, _sz = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_size_t

# name -> (restype, argtypes); mirrors include/wavemamba_hip.h one to one
SIGNATURES = {
    "wm_abi_version": (_i, []),
    "wm_build_id": (_c.c_char_p, []),
    "wm_strerror": (_c.c_char_p, [_i]),
    "wm_dwt2d_fwd": (_i, [_p] * 5 + [_i] * 5 + [_p]),
    "wm_dwt2d_bwd": (_i, [_p] * 5 + [_i] * 5 + [_p]),
    "wm_idwt2d_fwd": (_i, [_p] * 4 + [_i64] * 4 + [_p] + [_i] * 5 + [_p]),
    "wm_idwt2d_bwd": (_i, [_p] * 5 + [_i64] * 4 + [_i] * 5 + [_p]),
    "wm_selscan_fwd_workspace_bytes": (_sz, [_i] * 5),
    "wm_selscan_fwd": (_i, [_p] * 10 + [_p, _sz] + [_i] * 6 + [_p]),
    "wm_selscan_bwd_workspace_bytes": (_sz, [_i] * 5),
    "wm_selscan_bwd": (_i, [_p] * 15 + [_p, _sz] + [_i] * 6 + [_p]),
    "wm_ss2d_core_fwd_workspace_bytes": (_sz, [_i] * 7),
    "wm_ss2d_core_plan": (_i, [_i] * 6 + [_c.POINTER(_i)]),
    "wm_ss2d_core_fwd": (_i, [_p] * 10 + [_i, _p, _sz, _p] + [_i] * 7 + [_p]),
    "wm_ss2d_core_prep_bytes": (_sz, [_i]),
    "wm_ss2d_core_prep": (_i, [_p] * 6 + [_i] * 3 + [_p]),
    "wm_ss2d_core_bwd_workspace_bytes": (_sz, [_i] * 6),
    "wm_ss2d_core_bwd": (_i, [_p] * 16 + [_p, _sz] + [_i] * 6 + [_p]),
    "wm_lfss_in_fwd": (_i, [_p, _i, _p, _p, _c.c_float, _p, _p, _p, _i, _i64, _i, _i, _p]),
    "wm_lfss_mid_rz_fwd": (_i, [_p, _i, _i64, _p, _i, _p, _p, _c.c_float, _p, _p, _p, _c.c_float, _p, _p, _p, _p, _c.c_float, _p, _p,
                               _p, _p, _i, _i64, _i, _i, _p]),
    "wm_lfss_mid_fwd": (_i, [_p, _i, _i64, _p, _p, _i, _p, _p, _c.c_float, _p, _p, _p, _p, _c.c_float, _p, _p, _p, _p,
                             _i, _i64, _i, _i, _p]),
    "wm_lfss_out_fwd": (_i, [_p] * 6 + [_i, _i, _i64, _i, _i, _p]),
    "wm_lfss_out_conv_fwd": (_i, [_p] * 8 + [_i] * 6 + [_p]),
    "wm_layernorm2d_fwd": (_i, [_p, _p, _p, _c.c_float, _p, _i, _i64, _i, _p]),
    "wm_gram_workspace_bytes": (_sz, [_i, _i, _i64]),
    "wm_gram_fwd": (_i, [_p] * 6 + [_sz, _i, _i, _i64, _p]),
    "wm_dwconv3x3_wgrad": (_i, [_p] * 4 + [_i] * 4 + [_p]),
    "wm_layernorm2d_bwd": (_i, [_p, _p, _p, _c.c_float, _p, _p, _p, _i, _i64, _i, _p]),
    "wm_dwconv3x3_fwd": (_i, [_p] * 4 + [_i] * 6 + [_p]),
    "wm_layernorm_tok_fwd": (_i, [_p, _p, _p, _c.c_float, _p, _i64, _i, _p]),
    "wm_layernorm_tok_bwd": (_i, [_p, _p, _p, _c.c_float, _p, _p, _p, _i64, _i, _p]),
    "wm_image_pre_u8": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "wm_image_post_u8": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "wm_linear_wgrad": (_i, [_p, _p, _p, _i64, _i, _i, _p]),
    "wm_conv2d_wgrad_workspace_bytes": (_sz, [_i] * 6),
    "wm_conv2d_wgrad": (_i, [_p, _p, _p, _p, _p, _sz] + [_i] * 6 + [_p]),
    "wm_plane_sums": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "wm_l1_mean_fwd": (_i, [_p, _p, _p, _i64, _p]),
    "wm_l1_mean_bwd": (_i, [_p, _p, _p, _p, _i64, _p]),
    "wm_gate_fwd": (_i, [_p] * 3 + [_i, _i] + [_i64] * 4 + [_p]),
    "wm_gate_bwd": (_i, [_p] * 5 + [_i, _i] + [_i64] * 6 + [_p]),
    "wm_scale_add_fwd": (_i, [_p] * 4 + [_i, _i, _i64, _p]),
    "wm_scale_add_bwd": (_i, [_p] * 5 + [_i, _i, _i64, _p]),
    "wm_match_index": (_i, [_p, _p, _p, _p, _i, _i, _p]),
    "wm_attn_fold": (_i, [_p] * 6 + [_i, _i, _i, _p]),
    "wm_skff_workspace_bytes": (_sz, [_i, _i]),
    "wm_skff_fwd": (_i, [_p] * 7 + [_p, _sz] + [_i] * 5 + [_p]),
    "wm_conv2d_wfrag_bytes": (_sz, [_i, _i, _i]),
    "wm_conv2d_prep": (_i, [_p, _p, _i, _i, _i, _p]),
    "wm_conv2d_fwd": (_i, [_p] * 8 + [_i] * 8 + [_p]),
    "wm_conv2d_ln_fwd": (_i, [_p, _p, _p, _c.c_float, _p, _p, _p, _p] + [_i] * 5 + [_p]),
    "wm_patchify_conv_fwd": (_i, [_p] * 4 + [_i] * 6 + [_p]),
    "wm_conv2d_f16_steps": (_i, [_p] * 6 + [_i] * 7 + [_p]),
    "wm_conv2d_gated_fwd": (_i, [_p] * 7 + [_i] * 7 + [_p]),
    "wm_conv2d_select": (_i, [_i]),
    "wm_prof_enable": (None, [ctypes.c_uint]),
    "wm_prof_collect": (_i, [_c.POINTER(_i), _c.POINTER(_c.c_double)]),
    "wm_event_synchronize_relaxed": (_i, [_p]),
    "wm_zero_arena_register": (_i, [_p, _c.c_size_t]),
    "wm_zero_arena_unregister": (_i, [_p]),
}

_lib = None


class WaveMambaHipError(RuntimeError):
    pass


def load():
    """Load the HIP library (once).  Raises WaveMambaHipError if it is absent - never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise WaveMambaHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). The Wave-Mamba hot path has no CPU or PyTorch fallback.")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise WaveMambaHipError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise WaveMambaHipError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    # WAVEMAMBA_HIP_LIB may point at another BUILD of these sources (A/B timing of compile-time variants, tools/), never at
    # another ABI: a library with a different argument list would be called with shifted arguments (round 3 tolerated that
    # under WAVEMAMBA_HIP_AB=1; the mode is gone - ADVICE r3).
    if lib.wm_abi_version() != ABI_VERSION:
        raise WaveMambaHipError(f"ABI mismatch: library {lib.wm_abi_version()} != binding {ABI_VERSION}")
    if os.environ.get("WAVEMAMBA_HIP_LIB"):
        import sys
        print(f"[wave_mamba_amd] WAVEMAMBA_HIP_LIB: using {LIB_PATH} (build {lib.wm_build_id().decode()})", file=sys.stderr)
    _lib = lib
    return lib


def build_id():
    """The loaded library's source identity (build.py: source_id)."""
    return load().wm_build_id().decode()


def check(code, what):
    """Non-zero status -> RuntimeError (the reference raises RuntimeError on shape errors)."""
    if code != 0:
        msg = load().wm_strerror(code).decode()
        raise RuntimeError(f"{what} failed: {msg} (code {code})")
