"""
CompoundSpace (Pyrado/pyrado/spaces/compound.py:9-54): a uniformly chosen member space is sampled.
"""
from copy import deepcopy

import numpy as np

from .base import Space


class CompoundSpace(Space):
    def __init__(self, spaces):
        self._spaces = deepcopy(list(spaces))

    @property
    def shape(self):
        return tuple(s.shape for s in self._spaces)

    def _members(self):
        return tuple(self._spaces)

    def subspace(self, idcs):
        return self._spaces[idcs]

    def project_to(self, ele):
        raise NotImplementedError

    def contains(self, cand: np.ndarray, verbose: bool = False) -> bool:
        return any(s.contains(cand) for s in self._spaces)

    def sample_uniform(self, concrete_inf: float = 1e6) -> np.ndarray:
        return self._spaces[np.random.randint(len(self._spaces))].sample_uniform()
