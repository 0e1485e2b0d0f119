"""
RunningNormalizer (Pyrado/pyrado/utils/normalizing.py:7-84): maps data into [-1, 1] with the smallest and largest value
seen so far per entry.  NumPy arrays and torch tensors take the same path here; the reference spells the two out.
"""
import numpy as np
import torch as to

from .exceptions import ShapeErr, TypeErr


class RunningNormalizer:
    eps = 1e-3   # the span below which the bounds are pulled apart (1-norm over all entries, as the reference tests it)

    def __init__(self):
        self.reset()

    def reset(self):
        self._bound_lo = self._bound_up = None
        self._iter = 0

    def __repr__(self):
        return f'RunningNormalizer ID: {id(self)}\nbound_lo: {self._bound_lo}\nbound_up: {self._bound_up}\niter: {self._iter}'

    @staticmethod
    def _extremes(data):
        """ smallest / largest value per column, and the element-wise min / max of two such rows """
        if isinstance(data, np.ndarray):
            rows = np.atleast_2d(data)
            return rows.min(axis=0), rows.max(axis=0), np.fmin, np.fmax, lambda v: np.linalg.norm(v, ord=1)
        if isinstance(data, to.Tensor):
            rows = data.view(-1, 1) if data.ndim < 2 else data
            return rows.min(dim=0)[0], rows.max(dim=0)[0], to.min, to.max, lambda v: to.norm(v, p=1)
        raise TypeErr(given=data, expected_type=[np.ndarray, to.Tensor])

    def __call__(self, data):
        lo, up, smaller, larger, span = self._extremes(data)
        self._iter += 1
        if self._iter > 1:
            if self._bound_lo.shape != lo.shape:
                raise ShapeErr(given=lo, expected_match=self._bound_lo)
            lo, up = smaller(self._bound_lo, lo), larger(self._bound_up, up)
        if span(up - lo) < self.eps:   # a single sample, or a constant signal: the bounds must not collapse
            lo, up = lo - self.eps/2, up + self.eps/2
        self._bound_lo, self._bound_up = lo, up
        return (data - lo)/(up - lo)*2 - 1
