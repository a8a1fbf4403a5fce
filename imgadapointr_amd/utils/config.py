"""YAML -> attribute-dict configuration, schema-compatible with the reference.

Mirrors utils/config.py:18-58 of the reference: `cfg_from_yaml_file(path)` loads a YAML into
an EasyDict; a `_base_: other.yaml` key is REPLACED by the loaded sub-YAML (config.py:18-28),
so the reference's cfgs/*.yaml load unchanged.  `easydict` is not installed in this image, so
the small attribute-dict it provides is defined here under the same name.
"""
import os
import shutil

import yaml


class EasyDict(dict):
    """dict with attribute access; nested dicts (also inside lists/tuples) are converted."""

    def __init__(self, d=None, **kwargs):
        super().__init__()
        d = dict(d or {})
        d.update(kwargs)
        for k, v in d.items():
            self[k] = v

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __setattr__(self, k, v):
        self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __delattr__(self, k):
        try:
            del self[k]
        except KeyError:
            raise AttributeError(k)

    def update(self, *a, **kw):
        for k, v in dict(*a, **kw).items():
            self[k] = v

    def setdefault(self, k, default=None):
        if k not in self:
            self[k] = default
        return self[k]


def _load_yaml(path):
    with open(path, 'r') as f:
        return yaml.load(f, Loader=yaml.FullLoader)


def merge_new_config(config, new_config):
    for key, val in new_config.items():
        if not isinstance(val, dict):
            if key == '_base_':
                val = _load_yaml(new_config['_base_'])
                config[key] = EasyDict()
                merge_new_config(config[key], val)
            else:
                config[key] = val
            continue
        if key not in config:
            config[key] = EasyDict()
        merge_new_config(config[key], val)
    return config


def cfg_from_yaml_file(cfg_file):
    config = EasyDict()
    merge_new_config(config=config, new_config=_load_yaml(cfg_file))
    return config


def log_args_to_file(args, pre='args', logger=None):
    from .logger import print_log
    for key, val in args.__dict__.items():
        print_log(f'{pre}.{key} : {val}', logger=logger)


def log_config_to_file(cfg, pre='cfg', logger=None):
    from .logger import print_log
    for key, val in cfg.items():
        if isinstance(val, EasyDict):
            print_log(f'{pre}.{key} = edict()', logger=logger)
            log_config_to_file(val, pre=pre + '.' + key, logger=logger)
            continue
        print_log(f'{pre}.{key} : {val}', logger=logger)


def get_config(args, logger=None):
    from .logger import print_log
    if args.resume:
        cfg_path = os.path.join(args.experiment_path, 'config.yaml')
        if not os.path.exists(cfg_path):
            print_log('Failed to resume', logger=logger)
            raise FileNotFoundError(cfg_path)
        print_log(f'Resume yaml from {cfg_path}', logger=logger)
        args.config = cfg_path
    config = cfg_from_yaml_file(args.config)
    if not args.resume and args.local_rank == 0:
        save_experiment_config(args, config, logger)
    return config


def save_experiment_config(args, config, logger=None):
    from .logger import print_log
    config_path = os.path.join(args.experiment_path, 'config.yaml')
    shutil.copyfile(args.config, config_path)
    print_log(f'Copy the Config file from {args.config} to {config_path}', logger=logger)
