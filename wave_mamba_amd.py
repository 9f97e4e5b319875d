"""Alias so that `import wave_mamba_amd[.x.y]` resolves to the package directory `wave-mamba_amd/`.

The package directory carries the project's hyphenated name, which the `import` statement cannot
spell.  This module maps every `wave_mamba_amd...` name onto the SAME module objects as
`wave-mamba_amd...` (no second copy of any submodule is ever created).
"""
import importlib
import importlib.abc
import importlib.util
import os
import sys

_ALIAS, _REAL = "wave_mamba_amd", "wave-mamba_amd"
_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname == _ALIAS or fullname.startswith(_ALIAS + "."):
            return importlib.util.spec_from_loader(fullname, self)
        return None

    def create_module(self, spec):
        return importlib.import_module(_REAL + spec.name[len(_ALIAS):])

    def exec_module(self, module):
        pass


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
sys.modules[__name__] = importlib.import_module(_REAL)
