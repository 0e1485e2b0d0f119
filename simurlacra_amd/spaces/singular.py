"""
SingularStateSpace (Pyrado/pyrado/spaces/singular.py:7-22): a space holding one point; sampling returns that point.
"""
from typing import Sequence

import numpy as np

from .box import BoxSpace


class SingularStateSpace(BoxSpace):
    """ A box whose lower and upper bound coincide: the one state an environment always starts from. """

    def __init__(self, fixed_state: np.ndarray, labels: Sequence[str] = None):
        point = np.asarray(fixed_state)
        BoxSpace.__init__(self, point, point, labels=labels)
        self._fixed_state = fixed_state

    def sample_uniform(self, concrete_inf: float = 1e6) -> np.ndarray:
        # nothing to draw: a copy, so that the caller may write to it
        return np.array(self._fixed_state, copy=True)
