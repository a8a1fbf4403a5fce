"""Running last-value / mean trackers for one or several named quantities.

Call surface of utils/AverageMeter.py of the reference: AverageMeter(items=None) with
update(value | list[, n]), val(idx=None), count(idx=None), avg(idx=None), reset().
"""
import numpy as np


class AverageMeter(object):
    def __init__(self, items=None):
        self.items = items
        self.n_items = len(items) if items is not None else 1
        self.reset()

    def reset(self):
        self._stats = np.zeros((3, self.n_items), dtype=np.float64)  # rows: last value, sum, count

    def update(self, values, n=1):
        """`n` > 1: `values` are means over n samples (the runner reads its on-device loss accumulator once per log
        line); avg() stays the per-sample mean the reference computes with one update per iteration."""
        vals = values if isinstance(values, list) else [values]
        for i, v in enumerate(vals):
            self._stats[0, i] = v
            self._stats[1, i] += v * n
            self._stats[2, i] += n

    def _select(self, row, idx, as_int=False):
        conv = (lambda x: int(x)) if as_int else (lambda x: float(x))
        if idx is not None:
            return conv(row[idx])
        if self.items is None:
            return conv(row[0])
        return [conv(x) for x in row]

    def val(self, idx=None):
        return self._select(self._stats[0], idx)

    def count(self, idx=None):
        return self._select(self._stats[2], idx, as_int=True)

    def avg(self, idx=None):
        cnt = np.maximum(self._stats[2], 1)
        return self._select(self._stats[1] / cnt, idx)
