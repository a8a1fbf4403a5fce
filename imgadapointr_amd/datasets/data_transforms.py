"""Sample transforms (interface of datasets/data_transforms.py of the reference): `Compose` built
from [{'callback': name, 'parameters': {...}, 'objects': [...]}], with RandomSamplePoints :55-67
(random subset, zero padding to n_points), UpSamplePoints :69-90, RandomMirrorPoints :92-108 (one
random draw shared by all objects of a sample) and ToTensor :40-52."""
import numpy as np
import torch


class ToTensor(object):
    def __init__(self, parameters=None):
        pass

    def __call__(self, arr):
        if arr.ndim == 3:  # HWC image -> CHW
            arr = arr.transpose(2, 0, 1)
        return torch.from_numpy(np.ascontiguousarray(arr)).float()


class RandomSamplePoints(object):
    def __init__(self, parameters):
        self.n_points = parameters['n_points']

    def __call__(self, ptcloud):
        choice = np.random.permutation(ptcloud.shape[0])
        ptcloud = ptcloud[choice[:self.n_points]]
        if ptcloud.shape[0] < self.n_points:
            pad = np.zeros((self.n_points - ptcloud.shape[0], 3), dtype=ptcloud.dtype)
            ptcloud = np.concatenate([ptcloud, pad])
        return ptcloud


class UpSamplePoints(object):
    def __init__(self, parameters):
        self.n_points = parameters['n_points']

    def __call__(self, ptcloud):
        curr = ptcloud.shape[0]
        need = self.n_points - curr
        if need < 0:
            return ptcloud[np.random.permutation(self.n_points)]
        while curr <= need:
            ptcloud = np.tile(ptcloud, (2, 1))
            need -= curr
            curr *= 2
        return np.concatenate((ptcloud, ptcloud[np.random.permutation(need)]))


class RandomMirrorPoints(object):
    """rnd <= .25: mirror x and z; (.25,.5]: mirror x; (.5,.75]: mirror z; else: unchanged."""

    def __init__(self, parameters=None):
        pass

    def __call__(self, ptcloud, rnd_value):
        sx = -1.0 if rnd_value <= 0.5 else 1.0
        sz = -1.0 if (rnd_value <= 0.25 or 0.5 < rnd_value <= 0.75) else 1.0
        # `+ 0.0`: the reference multiplies by a 3x3 matrix (data_transforms.py:106), whose dot products turn the -0.0 of a
        # mirrored zero-padding row back into +0.0; everything else is unchanged
        ptcloud[:, :3] = ptcloud[:, :3] * np.array([sx, 1.0, sz], dtype=ptcloud.dtype) + 0.0
        return ptcloud


_CALLBACKS = {c.__name__: c for c in (ToTensor, RandomSamplePoints, UpSamplePoints, RandomMirrorPoints)}


class Compose(object):
    def __init__(self, transforms):
        self.transformers = [{'callback': _CALLBACKS[tr['callback']](tr.get('parameters')), 'objects': tr['objects']}
                             for tr in transforms]

    def __call__(self, data):
        for tr in self.transformers:
            transform, objects = tr['callback'], tr['objects']
            rnd_value = np.random.uniform(0, 1)
            for k, v in data.items():
                if k in objects:
                    data[k] = transform(v, rnd_value) if isinstance(transform, RandomMirrorPoints) else transform(v)
        return data
