"""Image-fused AdaPoinTr with the cyclic coarse-loss weight ("ImgAdaPoinTr" of the README).

Interface of models/internal/ImgResNetEncAdaPoinTrVariableLoss.py of the reference: registered
name, `forward(xyz, img)`, `get_loss(ret, gt, epoch)` with
`loss_recon = loss_coarse * alpha_loss[epoch] + loss_fine`, alpha = CycleLR(step 5, max 1.0,
base 0.01, gamma 0.995) tabulated for epochs 5..599 (:36-37,73,95).  The reference indexes that
595-entry table with the raw epoch (0..600) and raises IndexError from epoch 595 on; here the
index is clamped to the last entry.
"""
from .AdaPoinTr import AdaPoinTr
from .base_blocks import CycleLR
from .build import MODELS
from .pctransformer import PCTransformer

STEP_SIZE = 5
scheduler_loss = CycleLR(step_size=STEP_SIZE, max_lr=1.0, base_lr=0.01, gamma=0.995)


class PCTransformerImgOnly(PCTransformer):
    def __init__(self, config):
        config.use_img_features = True
        config.use_seg_features = False
        config.enable_denoising = True
        config.enable_segmentation_enhancement = False
        config.seg_before_enc = False
        config.seg_before_dec = False
        super().__init__(config)


@MODELS.register_module()
class ImgResNetEncAdaPoinTrVariableLoss(AdaPoinTr):
    def __init__(self, config, **kwargs):
        super().__init__(config, **kwargs)
        self.alpha_loss = [float(scheduler_loss.get_lr(last_epoch=e)) for e in range(STEP_SIZE, 600)]

    def build_trunk(self, config):
        return PCTransformerImgOnly(config)

    def coarse_loss_weight(self, epoch):
        return self.alpha_loss[min(int(epoch), len(self.alpha_loss) - 1)]

    def run_trunk(self, xyz, extra, denoise_noise):
        (img,) = extra
        return self.base_model(xyz, img, denoise_noise=denoise_noise)
