"""PCN-shaped synthetic dataset (no files): item = (taxonomy_id, model_id, (partial, gt)).

Same item schema as datasets/BasePCNDataset.py:85-113 of the reference for the PCN case
(2048-point partial with trailing zero padding, 16384-point ground truth).  Used by the bench,
the smoke test and the runner tests; the real PCN / ImgPCN file readers are SURVEY §8(f) row 3.
"""
import numpy as np
import torch
import torch.utils.data as data

from .build import DATASETS


@DATASETS.register_module()
class SyntheticPCN(data.Dataset):
    def __init__(self, config):
        self.npoints = int(config.get('N_POINTS', 16384))
        self.npartial = int(config.get('N_PARTIAL', 2048))
        self.subset = config.subset
        self.length = int(config.get('LENGTH', 64 if self.subset == 'train' else 8))
        self.seed = int(config.get('SEED', 0)) + {'train': 0, 'val': 1, 'test': 2}.get(self.subset, 3)
        self.taxonomies = ['02691156', '02933112', '02958343', '03001627']

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        rng = np.random.default_rng(self.seed * 1000003 + idx)
        # a noisy sphere shell as "object"; the partial sees one hemisphere, zero-padded to npartial
        v = rng.standard_normal((self.npoints, 3)).astype(np.float32)
        v /= np.linalg.norm(v, axis=-1, keepdims=True) + 1e-9
        gt = v * np.float32(0.3 + 0.1 * rng.random()) + rng.standard_normal((self.npoints, 3)).astype(np.float32) * 0.005
        vis = gt[gt[:, 0] > 0.0]
        n_vis = min(len(vis), self.npartial - self.npartial // 8)
        partial = np.zeros((self.npartial, 3), np.float32)
        partial[:n_vis] = vis[rng.permutation(len(vis))[:n_vis]]
        tax = self.taxonomies[idx % len(self.taxonomies)]
        return tax, f'synthetic-{idx:05d}', (torch.from_numpy(partial), torch.from_numpy(gt.astype(np.float32)))


@DATASETS.register_module()
class SyntheticImgPCN(SyntheticPCN):
    """ImgPCN-shaped items: (partial, gt, img[3,224,224]) — datasets/BasePCNDataset.py:100-102."""

    def __getitem__(self, idx):
        tax, mid, (partial, gt) = super().__getitem__(idx)
        rng = np.random.default_rng(self.seed * 7919 + idx)
        img = torch.from_numpy(rng.random((3, 224, 224), dtype=np.float32))
        return tax, mid, (partial, gt, img)


@DATASETS.register_module()
class SyntheticSegImgPCN(SyntheticImgPCN):
    """SegImgPCN-shaped items: (partial, gt, img, cls_vec[16]) — one-hot over the 16 ShapeNetPart
    classes, only classes 0/4/8/15 occur (datasets/SegImgPCNDataset.py:5-12 of the reference)."""

    def __getitem__(self, idx):
        tax, mid, (partial, gt, img) = super().__getitem__(idx)
        cls_vec = torch.zeros(16)
        cls_vec[(0, 4, 8, 15)[idx % 4]] = 1.0
        return tax, mid, (partial, gt, img, cls_vec)
