"""
SingularStateSpace (Pyrado/pyrado/spaces/singular.py:7-22): a space holding one point; sampling returns that point.
"""
from typing import Sequence

import numpy as np

from .box import BoxSpace


class SingularStateSpace(BoxSpace):
    def __init__(self, fixed_state: np.ndarray, labels: Sequence[str] = None):
        super().__init__(fixed_state, fixed_state, labels=labels)
        self._fixed_state = fixed_state

    def sample_uniform(self, concrete_inf: float = 1e6) -> np.ndarray:
        return self._fixed_state.copy()
