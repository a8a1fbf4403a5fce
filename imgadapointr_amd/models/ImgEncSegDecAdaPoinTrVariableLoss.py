"""Image-fused encoder + segmentation-fused decoder ("ImgEncSegDecAPTr" of the README).

Interface of models/internal/ImgEncSegDecAdaPoinTrVariableLoss.py:22-146 of the reference:
`forward(xyz, img, cls_vec)`, `get_loss(ret, gt, epoch)` with the cyclic coarse-loss weight.
The GDANet weights are loaded and frozen by the runner (tools/runner.py:55-66 of the reference).
"""
from .build import MODELS
from .ImgResNetEncAdaPoinTrVariableLoss import ImgResNetEncAdaPoinTrVariableLoss
from .pctransformer import PCTransformer


class PCTransformerImgSeg(PCTransformer):
    def __init__(self, config):
        config.use_img_features = True
        config.use_seg_features = True
        config.enable_denoising = True
        config.enable_segmentation_enhancement = False
        config.seg_before_enc = False
        config.seg_before_dec = True
        super().__init__(config)


@MODELS.register_module()
class ImgEncSegDecAdaPoinTrVariableLoss(ImgResNetEncAdaPoinTrVariableLoss):
    def build_trunk(self, config):
        return PCTransformerImgSeg(config)

    def run_trunk(self, xyz, extra, denoise_noise):
        img, cls_vec = extra
        return self.base_model(xyz, img, cls_vec, denoise_noise=denoise_noise)
