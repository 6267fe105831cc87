"""Minimal stand-in for pyhocon's ConfigTree (the getters the model constructor uses,
reference: code/model/network.py:481-499).  A real pyhocon ConfigTree works as well --
the model only calls get_int / get_float / get_bool / get_string / get_list / get_config."""


class ConfigTree(dict):
    _MISSING = object()

    @classmethod
    def from_dict(cls, d):
        t = cls()
        for k, v in d.items():
            t[k] = cls.from_dict(v) if isinstance(v, dict) else v
        return t

    def _get(self, key, default):
        if key in self:
            return self[key]
        if default is ConfigTree._MISSING:
            raise KeyError(key)
        return default

    def get_int(self, key, default=_MISSING):
        return int(self._get(key, default))

    def get_float(self, key, default=_MISSING):
        return float(self._get(key, default))

    def get_bool(self, key, default=_MISSING):
        return bool(self._get(key, default))

    def get_string(self, key, default=_MISSING):
        return str(self._get(key, default))

    def get_list(self, key, default=_MISSING):
        return list(self._get(key, default))

    def get_config(self, key, default=_MISSING):
        v = self._get(key, default)
        return v if hasattr(v, 'get_int') else ConfigTree.from_dict(v)
