"""Evaluation metrics (interface of utils/metrics.py of the reference): F-Score@1%, CDL1, CDL2
(x1000, `ignore_zeros=True`), EMDistance.

F-Score: the reference round-trips every sample through open3d on the host
(metrics.py:68-99).  open3d's compute_point_cloud_distance is the exact nearest-neighbour
distance, which is sqrt of the Chamfer kernel's dist1/dist2, so precision / recall are computed
on the device from one Chamfer launch per sample.
"""
import logging

import torch

from ..extensions.chamfer_dist import ChamferDistanceL1, ChamferDistanceL2, ChamferFunction


class Metrics(object):
    ITEMS = [{
        'name': 'F-Score', 'enabled': True, 'eval_func': '_get_f_score',
        'is_greater_better': True, 'init_value': 0
    }, {
        'name': 'CDL1', 'enabled': True, 'eval_func': '_get_chamfer_distancel1',
        'eval_object': ChamferDistanceL1(ignore_zeros=True), 'is_greater_better': False, 'init_value': 32767
    }, {
        'name': 'CDL2', 'enabled': True, 'eval_func': '_get_chamfer_distancel2',
        'eval_object': ChamferDistanceL2(ignore_zeros=True), 'is_greater_better': False, 'init_value': 32767
    }, {
        'name': 'EMDistance', 'enabled': True, 'eval_func': '_get_emd_distance',
        'is_greater_better': False, 'init_value': 32767
    }]

    @classmethod
    def get(cls, pred, gt, require_emd=False):
        values = []
        for item in cls.items():
            if not require_emd and 'emd' in item['eval_func']:
                values.append(torch.tensor(0.).to(gt.device))
            else:
                values.append(getattr(cls, item['eval_func'])(pred, gt))
        return values

    @classmethod
    def items(cls):
        return [i for i in cls.ITEMS if i['enabled']]

    @classmethod
    def names(cls):
        return [i['name'] for i in cls.items()]

    @classmethod
    def _get_f_score(cls, pred, gt, th=0.01):
        """Mean over the batch of 2PR/(P+R), P = share of pred points within th of gt, R vice versa."""
        assert pred.size(0) == gt.size(0)
        with torch.no_grad():
            dist1, dist2 = ChamferFunction.apply(pred.contiguous(), gt.contiguous())
            precision = (torch.sqrt(dist1) < th).float().mean(dim=1)
            recall = (torch.sqrt(dist2) < th).float().mean(dim=1)
            denom = precision + recall
            f = torch.where(denom > 0, 2 * recall * precision / denom.clamp_min(1e-30), torch.zeros_like(denom))
            return f.mean()

    @classmethod
    def _get_chamfer_distancel1(cls, pred, gt):
        return cls.ITEMS[1]['eval_object'](pred, gt) * 1000

    @classmethod
    def _get_chamfer_distancel2(cls, pred, gt):
        return cls.ITEMS[2]['eval_object'](pred, gt) * 1000

    @classmethod
    def _get_emd_distance(cls, pred, gt, eps=0.005, iterations=100):
        from ..extensions.emd import emd_module
        dist, _ = emd_module.emdModule()(pred, gt, eps, iterations)
        return torch.mean(torch.sqrt(dist)) * 1000

    def __init__(self, metric_name, values):
        self._items = Metrics.items()
        self._values = [item['init_value'] for item in self._items]
        self.metric_name = metric_name
        if isinstance(values, list):
            self._values = values
        elif isinstance(values, dict):
            index = {item['name']: i for i, item in enumerate(self._items)}
            for k, v in values.items():
                if k not in index:
                    logging.warning('Ignore Metric[Name=%s] due to disability.' % k)
                    continue
                self._values[index[k]] = v
        else:
            raise Exception('Unsupported value type: %s' % type(values))

    def state_dict(self):
        return {self._items[i]['name']: self._values[i] for i in range(len(self._items))}

    def __repr__(self):
        return str(self.state_dict())

    def better_than(self, other):
        if other is None:
            return True
        for i, item in enumerate(self._items):
            if item['name'] == self.metric_name:
                break
        else:
            raise Exception('Invalid metric name to compare.')
        a, b = self._values[i], other._values[i]
        return a > b if item['is_greater_better'] else a < b
