from .build import DATASETS, build_dataset_from_cfg  # noqa: F401
from . import SyntheticPCN  # noqa: F401
from . import BasePCNDataset  # noqa: F401  (registers PCN, PCNv2, ImgPCN, SegImgPCN)
