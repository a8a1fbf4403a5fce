"""MODELS registry and `build_model_from_cfg` (models/build.py:1-15 of the reference)."""
from ..utils import registry

MODELS = registry.Registry('models')


def build_model_from_cfg(cfg, **kwargs):
    """cfg.NAME selects the registered class, which is constructed as `cls(cfg)`."""
    return MODELS.build(cfg, **kwargs)
