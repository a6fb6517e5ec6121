"""Registry / build_from_cfg with the surface of mmcv.utils.Registry that the reference's builders use
(mmdet/models/builder.py:1-67, mmdet/core/bbox/builder.py:1-20): both ``@R.register_module()`` and the
bare ``@R.register_module`` decorator forms (offset_head_expand_feature.py:25 uses the bare one),
``build_from_cfg(cfg, registry, default_args)`` popping ``type``.
"""
import inspect


class Registry:
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    name = property(lambda self: self._name)
    module_dict = property(lambda self: self._module_dict)

    def __contains__(self, key):
        return key in self._module_dict

    def __repr__(self):
        return f'Registry(name={self._name}, items={sorted(self._module_dict)})'

    def get(self, key):
        return self._module_dict.get(key)

    def _add(self, cls, name=None, force=False):
        if not inspect.isclass(cls):
            raise TypeError(f'module must be a class, got {type(cls)}')
        key = name or cls.__name__
        if key in self._module_dict and not force:
            raise KeyError(f'{key} is already registered in {self._name}')
        self._module_dict[key] = cls

    def register_module(self, name=None, force=False, module=None):
        if inspect.isclass(name):          # bare decorator:  @R.register_module
            self._add(name, force=force)
            return name
        if module is not None:
            self._add(module, name, force)
            return module

        def _decorate(cls):
            self._add(cls, name, force)
            return cls
        return _decorate


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict) or 'type' not in cfg:
        raise TypeError(f'cfg must be a dict containing the key "type", got {cfg!r}')
    args = dict(cfg)
    for k, v in (default_args or {}).items():
        args.setdefault(k, v)
    obj_type = args.pop('type')
    if isinstance(obj_type, str):
        cls = registry.get(obj_type)
        if cls is None:
            raise KeyError(f'{obj_type} is not in the {registry.name} registry')
    elif inspect.isclass(obj_type):
        cls = obj_type
    else:
        raise TypeError(f'type must be a str or a class, got {type(obj_type)}')
    return cls(**args)
