"""Running value / sum / count meters (interface of utils/AverageMeter.py in the reference)."""


class AverageMeter(object):
    def __init__(self, items=None):
        self.items = items
        self.n_items = 1 if items is None else len(items)
        self.reset()

    def reset(self):
        self._val = [0] * self.n_items
        self._sum = [0] * self.n_items
        self._count = [0] * self.n_items

    def update(self, values):
        if isinstance(values, list):
            for i, v in enumerate(values):
                self._val[i] = v
                self._sum[i] += v
                self._count[i] += 1
        else:
            self._val[0] = values
            self._sum[0] += values
            self._count[0] += 1

    def _pick(self, arr, idx):
        if idx is not None:
            return arr[idx]
        return arr[0] if self.items is None else list(arr)

    def val(self, idx=None):
        return self._pick(self._val, idx)

    def count(self, idx=None):
        return self._pick(self._count, idx)

    def avg(self, idx=None):
        means = [s / c if c else 0.0 for s, c in zip(self._sum, self._count)]
        return self._pick(means, idx)
