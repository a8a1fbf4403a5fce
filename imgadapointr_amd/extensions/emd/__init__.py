from . import emd_module  # noqa: F401
