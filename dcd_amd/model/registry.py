"""Name -> constructor registries (DGDE/model/registry.py:3-5, DGDE/utils/registry.py:9-46)."""


class Registry(dict):
    """dict with a `register(name)` decorator / `register(name, obj)` call; duplicate names are an error."""

    def register(self, module_name, module=None):
        def _add(obj):
            assert module_name not in self, "%s already registered" % module_name
            self[module_name] = obj
            return obj
        if module is not None:
            _add(module)
            return None
        return _add


BACKBONES = Registry()
HEADS = Registry()
PREDICTOR = Registry()
