"""
Space interface (Pyrado/pyrado/spaces/base.py:14-169).
"""
from abc import ABC, abstractmethod
from copy import deepcopy
from functools import reduce

import numpy as np


class Space(ABC):
    bound_lo: np.ndarray
    bound_up: np.ndarray

    @property
    def bounds(self):
        return self.bound_lo, self.bound_up

    @property
    def bound_abs_up(self) -> np.ndarray:
        """ Element-wise larger magnitude of the two bounds. """
        return np.maximum(np.abs(self.bound_lo), np.abs(self.bound_up))

    @property
    def ele_dim(self) -> int:
        return self.bound_lo.shape[0]

    @property
    @abstractmethod
    def shape(self) -> tuple:
        raise NotImplementedError

    @property
    def flat_dim(self) -> int:
        return reduce(lambda a, b: a*b, self.shape)

    @property
    def labels(self):
        return None

    @abstractmethod
    def _members(self) -> tuple:
        raise NotImplementedError

    def __eq__(self, other):
        if type(other) is not type(self):
            return False
        return bool(np.all([np.all(a == b) for a, b in zip(self._members(), other._members())]))

    def copy(self):
        return deepcopy(self)

    @abstractmethod
    def contains(self, cand: np.ndarray, verbose: bool = False) -> bool:
        raise NotImplementedError

    def __contains__(self, cand):
        return self.contains(cand)

    @abstractmethod
    def sample_uniform(self, concrete_inf: float = 1e6) -> np.ndarray:
        raise NotImplementedError

    @abstractmethod
    def project_to(self, ele: np.ndarray) -> np.ndarray:
        raise NotImplementedError
