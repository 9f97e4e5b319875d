"""Name -> class registry with the surface of basicsr.utils.registry.Registry
(/root/reference/basicsr/utils/registry.py:4-82): register() as decorator or call, get(), `in`,
iteration, keys(), and the uniqueness assertion of _do_register (:38-41).

This is the registry of the stand-alone package.  Inside a basicsr tree the arch file binds to basicsr's own
ARCH_REGISTRY (see archs/wavemamba_arch.py and INTEGRATION.md).
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


# The package keeps its OWN registry.  It must not bind to basicsr's: when wavemamba_arch.py is dropped into
# basicsr/archs/, that copy registers `WaveMamba` in basicsr's ARCH_REGISTRY itself, and importing this package from it
# (for the operators) would otherwise register the package's class under the same name first - the uniqueness assertion
# of basicsr's Registry._do_register (:38-41) then fails (tests/test_dropin_basicsr.py).
ARCH_REGISTRY = Registry("arch")


def build_network(opt):
    """basicsr.archs.build_network (archs/__init__.py:19-25): pops 'type', instantiates with the rest."""
    opt = dict(opt)
    network_type = opt.pop("type")
    return ARCH_REGISTRY.get(network_type)(**opt)
