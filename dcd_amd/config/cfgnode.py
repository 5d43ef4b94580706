"""A small attribute-dict config node with the subset of the yacs API the reference uses
(`CfgNode`, `merge_from_file`, `merge_from_list`, `freeze`, `defrost`, `clone`); yacs itself is not
installed in this image.  Reference usage: DGDE/config/defaults.py:3-9, DGDE/tools/plain_train_net.py:116-137.
"""
import ast
import copy

import yaml


class CfgNode(dict):
    _FROZEN = "__frozen__"

    def __init__(self, init=None):
        super().__init__()
        object.__setattr__(self, CfgNode._FROZEN, False)
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        if object.__getattribute__(self, CfgNode._FROZEN):
            raise AttributeError("Attempted to set %s on a frozen CfgNode" % name)
        self[name] = value

    def is_frozen(self):
        return object.__getattribute__(self, CfgNode._FROZEN)

    def _set_frozen(self, flag):
        object.__setattr__(self, CfgNode._FROZEN, flag)
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(flag)

    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CfgNode()
        for k, v in self.items():
            out[k] = copy.deepcopy(v, memo)
        return out

    @staticmethod
    def _coerce(value, like):
        """YAML strings such as '("Car",)' are literal-evaluated, as yacs does."""
        if isinstance(value, str):
            try:
                value = ast.literal_eval(value)
            except (ValueError, SyntaxError):
                pass
        if isinstance(like, tuple) and isinstance(value, list):
            value = tuple(value)
        if isinstance(like, list) and isinstance(value, tuple):
            value = list(value)
        if isinstance(like, float) and isinstance(value, int) and not isinstance(value, bool):
            value = float(value)
        return value

    def _merge(self, other, path=""):
        for k, v in other.items():
            if k not in self:
                raise KeyError("Non-existent config key: %s%s" % (path, k))
            if isinstance(self[k], CfgNode):
                if not isinstance(v, dict):
                    raise ValueError("%s%s must be a mapping" % (path, k))
                self[k]._merge(v, path + k + ".")
            else:
                self[k] = CfgNode._coerce(v, self[k])

    def merge_from_file(self, filename):
        with open(filename, "r") as f:
            self._merge(yaml.safe_load(f) or {})

    def merge_from_other_cfg(self, other):
        self._merge(other)

    def merge_from_list(self, opts):
        if len(opts) % 2:
            raise ValueError("Override list has odd length: %s" % (opts,))
        for full_key, v in zip(opts[0::2], opts[1::2]):
            node = self
            parts = full_key.split(".")
            for p in parts[:-1]:
                if p not in node:
                    raise KeyError("Non-existent config key: %s" % full_key)
                node = node[p]
            if parts[-1] not in node:
                raise KeyError("Non-existent config key: %s" % full_key)
            node[parts[-1]] = CfgNode._coerce(v, node[parts[-1]])
