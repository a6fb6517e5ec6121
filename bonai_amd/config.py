"""Minimal loader for the reference's python config files.

Mirrors the surface the LOFT entry points rely on (SURVEY.md section 8b "Config files"):
``_base_`` lists merged depth-first, ``_delete_=True`` to replace instead of merge, attribute and
``.get()`` access on every nested dict (reference call sites: models/detectors/two_stage.py:31-43,
models/dense_heads/rpn_head.py:109,132), and ``merge_from_dict`` for ``--options k=v``
(tools/train.py:53,72-73).
"""
import os
import runpy

DELETE_KEY = '_delete_'
BASE_KEY = '_base_'


class ConfigDict(dict):
    """dict with attribute access; missing keys raise AttributeError (so hasattr works)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        del self[name]

    def copy(self):
        return ConfigDict(dict.copy(self))


def _wrap(obj):
    if isinstance(obj, dict):
        return ConfigDict((k, _wrap(v)) for k, v in obj.items())
    if isinstance(obj, list):
        return [_wrap(v) for v in obj]
    if isinstance(obj, tuple):
        return tuple(_wrap(v) for v in obj)
    return obj


def _merge(child, base):
    """Return base updated by child (child wins); honours ``_delete_``."""
    out = dict(base)
    for k, v in child.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get(DELETE_KEY, False):
            out[k] = _merge(v, out[k])
        elif isinstance(v, dict):
            out[k] = {kk: vv for kk, vv in v.items() if kk != DELETE_KEY}
        else:
            out[k] = v
    return out


def _load(path):
    path = os.path.abspath(path)
    ns = runpy.run_path(path)
    cfg = {k: v for k, v in ns.items()
           if (k == BASE_KEY or not k.startswith('_')) and not callable(v) and type(v).__name__ != 'module'}
    bases = cfg.pop(BASE_KEY, [])
    if isinstance(bases, str):
        bases = [bases]
    merged = {}
    for b in bases:
        bcfg = _load(os.path.join(os.path.dirname(path), b))
        dup = set(merged) & set(bcfg)
        if dup:
            raise KeyError(f'duplicate keys {sorted(dup)} in bases of {path}')
        merged.update(bcfg)
    return _merge(cfg, merged)


class Config(ConfigDict):
    @staticmethod
    def fromfile(path):
        cfg = Config(_wrap(_load(path)))
        dict.__setitem__(cfg, 'filename', os.path.abspath(path))
        return cfg

    def merge_from_dict(self, options):
        """``{'a.b.c': v}`` style overrides."""
        for key, val in options.items():
            node = self
            parts = key.split('.')
            for p in parts[:-1]:
                if p not in node or not isinstance(node[p], dict):
                    node[p] = ConfigDict()
                node = node[p]
            node[parts[-1]] = _wrap(val)
