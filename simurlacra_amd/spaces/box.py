"""
BoxSpace (Pyrado/pyrado/spaces/box.py:10-171): inclusive bounds, NaN candidates raise, `project_to` returns the
SAME array object when it already lies inside (the caller's array is then what the simulation sees).
"""
from typing import Sequence

import numpy as np

from ..utils.exceptions import ShapeErr, TypeErr, ValueErr
from .base import Space


class BoxSpace(Space):
    def __init__(self, bound_lo, bound_up, shape=None, labels: Sequence[str] = None):
        if shape is not None:
            self.bound_lo = np.ones(shape)*bound_lo
            self.bound_up = np.ones(shape)*bound_up
        else:
            try:
                self.bound_lo = np.atleast_1d(np.array(bound_lo, dtype=np.float64))
                self.bound_up = np.atleast_1d(np.array(bound_up, dtype=np.float64))
            except (TypeError, ValueError):
                raise TypeErr(given=bound_lo, expected_type=[float, list, np.ndarray])
            if self.bound_lo.shape != self.bound_up.shape:
                raise ShapeErr(given=self.bound_lo, expected_match=self.bound_up)
        if labels is not None:
            labels = np.array(labels, dtype=object)
            if labels.shape != self.shape:
                raise ShapeErr(given=labels, expected_match=self)
            self._labels = labels
        else:
            self._labels = np.full(self.shape, None, dtype=object)

    def _members(self):
        return self.bound_lo, self.bound_up, self._labels

    @property
    def shape(self) -> tuple:
        return self.bound_lo.shape

    @property
    def labels(self) -> np.ndarray:
        return self._labels

    def subspace(self, idcs):
        mask = idcs if isinstance(idcs, np.ndarray) and idcs.dtype == np.bool_ else self.create_mask(idcs)
        return BoxSpace(np.atleast_1d(self.bound_lo[mask]), np.atleast_1d(self.bound_up[mask]),
                        labels=self._labels[mask])

    def create_mask(self, *idcs):
        mask = np.zeros(self.shape, dtype=np.bool_)
        if len(idcs) == 1 and hasattr(idcs[0], '__iter__') and not isinstance(idcs[0], str):
            idcs = idcs[0]
        for idx in idcs:
            if isinstance(idx, str):
                hits = np.flatnonzero(self._labels == idx)
                if hits.size == 0:
                    raise ValueErr(msg=f'Label {idx} not found in {self}')
                idx = int(hits[0])
            if np.all(mask[idx]):
                raise ValueErr(msg=f'Duplicate index {idx}')
            mask[idx] = True
        return mask

    def shrink(self, new_lo: np.ndarray, new_up: np.ndarray):
        if not (isinstance(new_lo, np.ndarray) and isinstance(new_up, np.ndarray)):
            raise TypeErr(given=new_lo, expected_type=np.ndarray)
        if new_lo.shape != new_up.shape:
            raise ShapeErr(given=new_up, expected_match=new_lo)
        if not (new_lo >= self.bound_lo).all():
            raise ValueErr(msg='At least one new lower bound is too low!')
        if not (new_up <= self.bound_up).all():
            raise ValueErr(msg='At least one new upper bound is too high!')
        res = self.copy()
        res.bound_lo, res.bound_up = new_lo, new_up
        return res

    def contains(self, cand: np.ndarray, verbose: bool = False) -> bool:
        if cand.shape != self.shape:
            raise ShapeErr(given=cand, expected_match=self)
        if np.isnan(cand).any():
            raise ValueErr(msg=f'At least one value is NaN! {cand}')
        ok = bool(np.all((cand >= self.bound_lo) & (cand <= self.bound_up)))
        if not ok and verbose:
            print(f'lower bound {self.bound_lo}\ncandidate   {cand}\nupper bound {self.bound_up}')
        return ok

    def sample_uniform(self, concrete_inf: float = 1e6) -> np.ndarray:
        lo, up = self.bound_lo.copy(), self.bound_up.copy()
        lo[lo == -np.inf] = -concrete_inf
        up[up == np.inf] = concrete_inf
        return np.random.uniform(lo, up)

    def project_to(self, ele: np.ndarray) -> np.ndarray:
        return ele if self.contains(ele) else np.clip(ele, self.bound_lo, self.bound_up)

    @staticmethod
    def cat(spaces):
        spaces = [s for s in spaces if s is not None]
        lo, up, labels = [], [], []
        for s in spaces:
            if not isinstance(s, BoxSpace):
                raise TypeErr(given=s, expected_type=BoxSpace)
            lo.extend(s.bound_lo)
            up.extend(s.bound_up)
            labels.extend(s.labels)
        return BoxSpace(lo, up, labels=labels)

    def __str__(self):
        return f'{type(self).__name__} lower {self.bound_lo} upper {self.bound_up}'
