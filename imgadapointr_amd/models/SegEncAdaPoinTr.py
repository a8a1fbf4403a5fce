"""Segmentation-fused encoder variant (interface of models/internal/SegEncAdaPoinTr.py:11-129 of
the reference): `forward(xyz, img, cls_vec)` (the image is accepted and ignored, as in the
reference :22-23), plain AdaPoinTr loss.  Its YAMLs use center_num [512, 512]: GDANet's
disentangling step needs at least 512 proxy points."""
from .AdaPoinTr import AdaPoinTr
from .build import MODELS
from .pctransformer import PCTransformer


class PCTransformerSegOnly(PCTransformer):
    def __init__(self, config):
        config.use_img_features = False
        config.use_seg_features = True
        config.enable_denoising = True
        config.enable_segmentation_enhancement = True
        config.seg_before_enc = True
        config.seg_before_dec = False
        super().__init__(config)


@MODELS.register_module()
class SegEncAdaPoinTr(AdaPoinTr):
    def build_trunk(self, config):
        return PCTransformerSegOnly(config)

    def run_trunk(self, xyz, extra, denoise_noise):
        _img, cls_vec = extra
        return self.base_model(xyz, None, cls_vec, denoise_noise=denoise_noise)
