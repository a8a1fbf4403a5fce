"""PCN-family datasets (interface of datasets/{BasePCNDataset,PCNDataset,ImgPCNDataset,
SegImgPCNDataset}.py of the reference): item = (taxonomy_id, model_id, (partial, gt[, img[, cls_vec]])).

File lists come from the category JSON (`[{taxonomy_id, train: [...], val: [...], test: [...]}]`),
8 partial renderings per object for training (one picked at random per access), images are one RGB
view resized to 224 (shorter side, like torchvision's Resize(224)) and scaled to [0,1].
"""
import json
import random

import numpy as np
import torch
import torch.utils.data as data

from ..utils.io_module import IO
from . import data_transforms
from .build import DATASETS

SEG_LIST = ['02691156', '03001627', '03636649', '04379243']
CLASSES = {'02691156': 0, '02773838': 1, '02954340': 2, '02958343': 3, '03001627': 4, '03261776': 5,
           '03467517': 6, '03624134': 7, '03636649': 8, '03642806': 9, '03790512': 10, '03797390': 11,
           '03948459': 12, '04099429': 13, '04225987': 14, '04379243': 15}


def load_image_224(path):
    from PIL import Image
    img = Image.open(path).convert('RGB')
    w, h = img.size
    # torchvision's Resize(224) on a PIL image: shorter side -> 224, longer side -> int(224 * long / short) (truncated),
    # untouched if the shorter side already is 224
    if min(w, h) != 224:
        if w <= h:
            img = img.resize((224, int(224 * h / w)), Image.BILINEAR)
        else:
            img = img.resize((int(224 * w / h), 224), Image.BILINEAR)
    return torch.from_numpy(np.asarray(img, dtype=np.float32) / 255.0).permute(2, 0, 1).contiguous()


class BasePCNDataset(data.Dataset):
    def __init__(self, config, include_images=False, num_imgs_per_obj=1, include_segmentation=False,
                 SEG_LIST=None, CLASSES=None):
        self.partial_points_path = config.PARTIAL_POINTS_PATH
        self.complete_points_path = config.COMPLETE_POINTS_PATH
        self.category_file = config.CATEGORY_FILE_PATH
        self.npoints = config.N_POINTS
        self.subset = config.subset
        self.cars = config.CARS
        self.include_images = include_images
        self.num_imgs_per_obj = num_imgs_per_obj
        self.include_segmentation = include_segmentation
        # `views` (dataset `others`, e.g. others: {subset: test, views: 12}): keep that many of the sampled views per
        # object and return them as [V,3,224,224] - the README's 6 / 12 "random viewing points" evaluation
        # (README.md:80-85); 1 = the released code's behaviour (one surviving view, BasePCNDataset.py:72-75)
        self.views = int(getattr(config, 'views', 1) or 1)
        if include_images:
            self.img_path = config.IMG_PATH
            self.img_idxs = list(range(24))
        self.SEG_LIST, self.CLASSES = SEG_LIST, CLASSES
        with open(self.category_file) as f:
            self.dataset_categories = json.loads(f.read())
        if self.cars:
            self.dataset_categories = [dc for dc in self.dataset_categories if dc['taxonomy_id'] == '02958343']
        self.n_renderings = 8 if self.subset == 'train' else 1
        self.file_list = self._get_file_list(self.subset, self.n_renderings)
        self.transforms = self._get_transforms(self.subset)

    def _get_transforms(self, subset):
        steps = [{'callback': 'RandomSamplePoints', 'parameters': {'n_points': 2048}, 'objects': ['partial']}]
        if subset == 'train':
            steps.append({'callback': 'RandomMirrorPoints', 'objects': ['partial', 'gt']})
        steps.append({'callback': 'ToTensor', 'objects': ['partial', 'gt']})
        return data_transforms.Compose(steps)

    def _fmt(self, pattern, *args):
        return pattern % args[:pattern.count('%')]

    def _get_file_list(self, subset, n_renderings=1):
        file_list = []
        for dc in self.dataset_categories:
            for s in dc[subset]:
                item = {
                    'taxonomy_id': dc['taxonomy_id'], 'model_id': s,
                    'partial_path': [self.partial_points_path % (subset, dc['taxonomy_id'], s, i)
                                     for i in range(n_renderings)],
                    'gt_path': self._fmt(self.complete_points_path, subset, dc['taxonomy_id'], s, s),
                }
                if self.include_images:
                    # the reference samples num_imgs_per_obj views but keeps only the last one
                    # (BasePCNDataset.py:72-75): one view per object survives
                    sample = random.sample(self.img_idxs, max(self.num_imgs_per_obj, self.views))
                    item['img_path'] = self._fmt(self.img_path, subset, dc['taxonomy_id'], s, sample[-1])
                    if self.views > 1:
                        item['img_paths'] = [self._fmt(self.img_path, subset, dc['taxonomy_id'], s, i)
                                             for i in sample[-self.views:]]
                if self.include_segmentation:
                    cls_vec = np.zeros(16)
                    if dc['taxonomy_id'] in self.SEG_LIST:
                        cls_vec[self.CLASSES[dc['taxonomy_id']]] = 1
                    item['cls_vec'] = cls_vec
                file_list.append(item)
        return file_list

    def __getitem__(self, idx):
        sample = self.file_list[idx]
        rand_idx = random.randint(0, self.n_renderings - 1) if self.subset == 'train' else 0
        item = {}
        for ri in ('partial', 'gt'):
            path = sample[f'{ri}_path']
            if isinstance(path, list):
                path = path[rand_idx]
            item[ri] = IO.get(path).astype(np.float32)
        if self.transforms is not None:
            item = self.transforms(item)
        out = (item['partial'], item['gt'])
        if self.include_images:
            if self.views > 1:
                out = out + (torch.stack([load_image_224(p) for p in sample['img_paths']]),)
            else:
                out = out + (load_image_224(sample['img_path']),)
        if 'cls_vec' in sample:
            out = out + (torch.from_numpy(sample['cls_vec']).float(),)
        return sample['taxonomy_id'], sample['model_id'], out

    def __len__(self):
        return len(self.file_list)


@DATASETS.register_module()
class PCN(BasePCNDataset):
    def __init__(self, config):
        super().__init__(config)


@DATASETS.register_module()
class PCNv2(BasePCNDataset):
    """PCN with partials up-sampled (tiled + random repeats) to 2048 instead of zero-padded
    (datasets/PCNDataset.py:25-40 of the reference)."""

    def __init__(self, config):
        super().__init__(config)

    def _get_transforms(self, subset):
        steps = [{'callback': 'UpSamplePoints', 'parameters': {'n_points': 2048}, 'objects': ['partial']}]
        if subset == 'train':
            steps.append({'callback': 'RandomMirrorPoints', 'objects': ['partial', 'gt']})
        steps.append({'callback': 'ToTensor', 'objects': ['partial', 'gt']})
        return data_transforms.Compose(steps)


@DATASETS.register_module()
class ImgPCN(BasePCNDataset):
    def __init__(self, config, num_imgs_per_obj=1):       # ImgPCNDataset.py:8 (SegImgPCN samples 6, SegImgPCNDataset.py:18)
        super().__init__(config, include_images=True, num_imgs_per_obj=num_imgs_per_obj)


@DATASETS.register_module()
class SegImgPCN(BasePCNDataset):
    def __init__(self, config, num_imgs_per_obj=6):
        super().__init__(config, include_images=True, num_imgs_per_obj=num_imgs_per_obj,
                         include_segmentation=True, SEG_LIST=SEG_LIST, CLASSES=CLASSES)
