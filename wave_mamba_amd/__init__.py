"""wave_mamba_amd: MI355X-native (gfx950) implementation of the Wave-Mamba hot path.

(`wave-mamba_amd/`, the project's hyphenated name, is a symlink to this directory.)

    ops.dwt_init / ops.iwt_init / ops.selective_scan_fn   hand-written HIP behind a C ABI
    archs.wavemamba_arch.WaveMamba                        the reference's registry entry, re-built
"""
from . import _lib, ops, registry, trainer, inference   # noqa: F401
from .registry import ARCH_REGISTRY, build_network   # noqa: F401
from .archs import wavemamba_arch           # noqa: F401  (registers 'WaveMamba')
from .archs.wavemamba_arch import WaveMamba  # noqa: F401

__version__ = "0.1.0"
