"""
Space interface (Pyrado/pyrado/spaces/base.py:14-169).
"""
from abc import ABC, abstractmethod
from copy import deepcopy

import numpy as np


class Space(ABC):
    """ A set of arrays with element-wise bounds.  Concrete spaces provide the shape, membership, uniform sampling and
    projection; the rest follows from the two bound arrays. """
    bound_lo: np.ndarray
    bound_up: np.ndarray

    # provided by the concrete space ---------------------------------------------------------------------------------
    @property
    @abstractmethod
    def shape(self) -> tuple:
        raise NotImplementedError

    @abstractmethod
    def _members(self) -> tuple:
        """ The arrays two spaces of one type are compared by. """
        raise NotImplementedError

    @abstractmethod
    def contains(self, cand: np.ndarray, verbose: bool = False) -> bool:
        raise NotImplementedError

    @abstractmethod
    def sample_uniform(self, concrete_inf: float = 1e6) -> np.ndarray:
        raise NotImplementedError

    @abstractmethod
    def project_to(self, ele: np.ndarray) -> np.ndarray:
        raise NotImplementedError

    # derived --------------------------------------------------------------------------------------------------------
    labels = property(lambda self: None)
    bounds = property(lambda self: (self.bound_lo, self.bound_up))
    ele_dim = property(lambda self: self.bound_lo.shape[0])

    @property
    def flat_dim(self) -> int:
        n = 1
        for extent in self.shape:
            n *= extent
        return n

    @property
    def bound_abs_up(self) -> np.ndarray:
        """ Element-wise larger magnitude of the two bounds. """
        return np.maximum(np.abs(self.bound_lo), np.abs(self.bound_up))

    def copy(self):
        return deepcopy(self)

    def __contains__(self, cand):
        return self.contains(cand)

    def __eq__(self, other):
        if type(other) is not type(self):
            return False
        return all(bool(np.all(mine == theirs)) for mine, theirs in zip(self._members(), other._members()))
