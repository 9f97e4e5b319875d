"""TEST INFRASTRUCTURE - not part of the product.

Runs the package's own WaveMamba module tree with a different hot-path operator set (normally the CPU oracle,
oracle/oracle.py) so that tests, __graft_entry__.smoke() and bench.py's cpu_baseline leg can check / time the same
network on host cores.  The product arch file (wave_mamba_amd/archs/wavemamba_arch.py) has no installer for this:
the one class attribute it reads its operators from is patched here, from outside the package.
"""
import contextlib


def _arch():
    from wave_mamba_amd.archs import wavemamba_arch
    return wavemamba_arch


def set_ops_backend(backend):
    """Install `backend` (an object exposing dwt_init, iwt_init_pair, selective_scan_fn, ...) as the operator set of
    the arch module; returns the previous one.  None restores the HIP operators."""
    arch = _arch()
    prev = arch._OpsBackend.impl
    arch._OpsBackend.impl = arch._hip_ops if backend is None else backend
    return prev


@contextlib.contextmanager
def ops_backend(backend):
    prev = set_ops_backend(backend)
    try:
        yield
    finally:
        set_ops_backend(prev)
