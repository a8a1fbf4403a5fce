"""DATASETS registry (datasets/build.py:1-18 of the reference)."""
from ..utils import registry

DATASETS = registry.Registry('dataset')


def build_dataset_from_cfg(cfg, default_args=None):
    """cfg = the dataset YAML (`_base_`), default_args = the `others` dict (subset, bs, ...)."""
    return DATASETS.build(cfg, default_args=default_args)
