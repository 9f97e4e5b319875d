"""Drop-in proof (build container only: needs /root/reference at run time, nothing of it is stored in the repo).

A temporary `basicsr/` package is assembled from the reference's REAL `utils/registry.py` and REAL `archs/__init__.py`
(copied into a temp dir at test time), a two-function stand-in for `basicsr.utils` (the real one imports cv2, which this
image lacks), and THIS repo's `archs/wavemamba_arch.py` copied into `basicsr/archs/`.  Then exactly what the reference's
callers do: the `*_arch.py` auto-scan (archs/__init__.py:12-16), `build_network({'type': 'WaveMamba', ...})`
(:19-25), `load_state_dict(torch.load(path)['params'], strict=False)` (inference_wavemamba.py:77), a forward.
"""
import importlib
import logging
import os
import shutil
import sys

import pytest
import torch

REF = "/root/reference/basicsr"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = dict(in_chn=3, wf=8, n_l_blocks=[1, 1, 1], n_h_blocks=[1, 1, 1], ffn_scale=2.0)

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree exists only in the build container")


@pytest.fixture()
def basicsr_tree(tmp_path):
    pkg = tmp_path / "basicsr"
    (pkg / "utils").mkdir(parents=True)
    (pkg / "archs").mkdir()
    (pkg / "__init__.py").write_text("")
    # stand-in for basicsr/utils/__init__.py: the two names archs/__init__.py imports (:5)
    (pkg / "utils" / "__init__.py").write_text(
        "import logging, os\n"
        "def get_root_logger(*a, **k):\n    return logging.getLogger('basicsr')\n"
        "def scandir(dir_path, suffix=None, recursive=False, full_path=False):\n"
        "    for e in os.scandir(dir_path):\n"
        "        if e.is_file() and not e.name.startswith('.'):\n            yield e.name\n")
    shutil.copy(os.path.join(REF, "utils", "registry.py"), pkg / "utils" / "registry.py")      # the real registry
    shutil.copy(os.path.join(REF, "archs", "__init__.py"), pkg / "archs" / "__init__.py")      # the real auto-scan
    shutil.copy(os.path.join(ROOT, "wave_mamba_amd", "archs", "wavemamba_arch.py"), pkg / "archs" / "wavemamba_arch.py")
    saved = {k: v for k, v in sys.modules.items() if k == "basicsr" or k.startswith("basicsr.")}
    for k in saved:
        del sys.modules[k]
    sys.path.insert(0, str(tmp_path))
    importlib.invalidate_caches()
    try:
        yield tmp_path
    finally:
        sys.path.remove(str(tmp_path))
        for k in [k for k in sys.modules if k == "basicsr" or k.startswith("basicsr.")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_arch_file_drops_into_basicsr(basicsr_tree, tmp_path):
    import wave_mamba_amd as wm                                   # the package is installed next to basicsr
    archs = importlib.import_module("basicsr.archs")              # runs the *_arch.py auto-scan
    reg = importlib.import_module("basicsr.utils.registry").ARCH_REGISTRY
    assert "WaveMamba" in reg.keys()
    dropped = sys.modules["basicsr.archs.wavemamba_arch"]
    assert reg.get("WaveMamba") is dropped.WaveMamba
    assert dropped.WaveMamba is not wm.WaveMamba                  # two module objects, two registries, no name clash
    assert dropped._hip_ops is wm.ops                             # operators come from the installed package

    torch.manual_seed(0)
    net = archs.build_network(dict(type="WaveMamba", **CFG))      # archs/__init__.py:19-25
    assert isinstance(net, torch.nn.Module) and hasattr(net, "restoration_network")
    for name in ("forward", "test", "test_tile", "check_image_size", "encode_and_decode", "print_network"):
        assert callable(getattr(net, name)), name

    # a reference-format checkpoint written by the package's trainer loads the way inference_wavemamba.py:77 does
    torch.manual_seed(1)
    donor = wm.WaveMamba(**CFG)
    ckpt = tmp_path / "net_g.pth"
    wm.trainer.save_network(donor, str(ckpt))
    net.load_state_dict(torch.load(str(ckpt), weights_only=True)["params"], strict=False)
    for (k1, v1), (k2, v2) in zip(net.state_dict().items(), donor.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2), k1

    # forward through the dropped-in module tree (CPU container: the oracle stands in for the HIP operators)
    from oracle import oracle
    x = torch.rand(1, 3, 32, 48, generator=torch.Generator().manual_seed(3))
    prev, dropped._OpsBackend.impl = dropped._OpsBackend.impl, oracle
    try:
        with torch.no_grad():
            y = net.eval().test(x)
    finally:
        dropped._OpsBackend.impl = prev
    from oracle import backend as oracle_backend
    with oracle_backend.ops_backend(oracle), torch.no_grad():
        y_pkg = donor.eval()(x)
    assert y.shape == x.shape and torch.equal(y, y_pkg)


def test_registering_twice_is_rejected_like_the_reference(basicsr_tree):
    importlib.import_module("basicsr.archs")
    reg = importlib.import_module("basicsr.utils.registry").ARCH_REGISTRY
    with pytest.raises(AssertionError):
        reg.register(sys.modules["basicsr.archs.wavemamba_arch"].WaveMamba)      # registry.py:38-41
