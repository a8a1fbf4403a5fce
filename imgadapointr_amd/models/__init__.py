from .build import MODELS, build_model_from_cfg  # noqa: F401
from . import AdaPoinTr  # noqa: F401  (registers 'AdaPoinTr')
from . import ImgResNetEncAdaPoinTrVariableLoss  # noqa: F401
from . import ImgEncSegDecAdaPoinTrVariableLoss  # noqa: F401
from . import SegEncAdaPoinTr  # noqa: F401
from . import chamfer_baselines  # noqa: F401  (registers PCN, FoldingNet, TopNet)
