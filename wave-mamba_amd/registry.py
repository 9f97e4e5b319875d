"""Name -> class registry with the surface of basicsr.utils.registry.Registry
(/root/reference/basicsr/utils/registry.py:4-82): register() as decorator or call, get(), `in`,
iteration, keys(), and the uniqueness assertion of _do_register (:38-41).

When the user's `basicsr` package is importable (the arch file dropped into basicsr/archs/), the
real basicsr ARCH_REGISTRY is used instead, so `build_network({'type': 'WaveMamba', ...})` finds it.
"""


class Registry:
    def __init__(self, name):
        self._name = name
        self._obj_map = {}

    def _do_register(self, name, obj):
        assert name not in self._obj_map, (
            f"An object named '{name}' was already registered in '{self._name}' registry!")
        self._obj_map[name] = obj

    def register(self, obj=None):
        if obj is None:                      # decorator form
            def deco(target):
                self._do_register(target.__name__, target)
                return target
            return deco
        self._do_register(obj.__name__, obj)

    def get(self, name):
        found = self._obj_map.get(name)
        if found is None:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
        return found

    def __contains__(self, name):
        return name in self._obj_map

    def __iter__(self):
        return iter(self._obj_map.items())

    def keys(self):
        return self._obj_map.keys()


def _resolve_arch_registry():
    try:                                     # dropped into a real basicsr tree
        from basicsr.utils.registry import ARCH_REGISTRY as reg
        return reg
    except Exception:
        return Registry("arch")


ARCH_REGISTRY = _resolve_arch_registry()


def build_network(opt):
    """basicsr.archs.build_network (archs/__init__.py:19-25): pops 'type', instantiates with the rest."""
    opt = dict(opt)
    network_type = opt.pop("type")
    return ARCH_REGISTRY.get(network_type)(**opt)
