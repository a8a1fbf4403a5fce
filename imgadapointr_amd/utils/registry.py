"""Name -> class registry with the call surface the reference's builders rely on.

Mirrors utils/registry.py:194-290 of the reference as far as the hot path uses it:
`@REG.register_module()` (optionally `name=`, `force=`), `REG.get(name)`, `REG.build(cfg)`,
and `build_from_cfg(cfg, registry, default_args)` which looks up `cfg['NAME']` and calls
`cls(cfg)` — raising KeyError for an unregistered name exactly as the reference does
(registry.py:278-280).  Scopes / child registries of the mmcv original are not needed by any
caller on the path and are left out.
"""
import inspect

from . import config as _config


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict):
        raise TypeError(f'cfg must be a dict, but got {type(cfg)}')
    if 'NAME' not in cfg:
        if default_args is None or 'NAME' not in default_args:
            raise KeyError('`cfg` or `default_args` must contain the key "NAME", '
                           f'but got {cfg}\n{default_args}')
    if not isinstance(registry, Registry):
        raise TypeError(f'registry must be a Registry object, but got {type(registry)}')
    if not (isinstance(default_args, dict) or default_args is None):
        raise TypeError(f'default_args must be a dict or None, but got {type(default_args)}')
    if default_args is not None:
        cfg = _config.merge_new_config(cfg, default_args)
    obj_type = cfg.get('NAME')
    if isinstance(obj_type, str):
        obj_cls = registry.get(obj_type)
        if obj_cls is None:
            raise KeyError(f'{obj_type} is not in the {registry.name} registry')
    elif inspect.isclass(obj_type):
        obj_cls = obj_type
    else:
        raise TypeError(f'type must be a str or valid type, but got {type(obj_type)}')
    try:
        return obj_cls(cfg)
    except Exception as e:  # keep the class name in the message, like the reference
        raise type(e)(f'{obj_cls.__name__}: {e}')


class Registry:
    def __init__(self, name, build_func=None):
        self._name = name
        self._module_dict = {}
        self.build_func = build_func or build_from_cfg

    def __len__(self):
        return len(self._module_dict)

    def __contains__(self, key):
        return self.get(key) is not None

    def __repr__(self):
        return f'{type(self).__name__}(name={self._name}, items={self._module_dict})'

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key)

    def build(self, *args, **kwargs):
        return self.build_func(*args, **kwargs, registry=self)

    def _register_module(self, module_class, module_name=None, force=False):
        if not inspect.isclass(module_class):
            raise TypeError(f'module must be a class, but got {type(module_class)}')
        names = module_name if module_name is not None else module_class.__name__
        if isinstance(names, str):
            names = [names]
        for n in names:
            if not force and n in self._module_dict:
                raise KeyError(f'{n} is already registered in {self.name}')
            self._module_dict[n] = module_class

    def register_module(self, name=None, force=False, module=None):
        if not isinstance(force, bool):
            raise TypeError(f'force must be a boolean, but got {type(force)}')
        if not (name is None or isinstance(name, str) or
                (isinstance(name, (list, tuple)) and all(isinstance(n, str) for n in name))):
            raise TypeError(f'name must be None, a str or a sequence of str, but got {type(name)}')
        if module is not None:
            self._register_module(module_class=module, module_name=name, force=force)
            return module

        def _register(cls):
            self._register_module(module_class=cls, module_name=name, force=force)
            return cls

        return _register
