"""
ActNormWrapper (Pyrado/pyrado/environment_wrappers/action_normalization.py:7-22): actions in [-1, 1] are mapped
affinely onto the wrapped environment's current action bounds.
"""
import numpy as np

from ..spaces.box import BoxSpace
from .base import EnvWrapperAct


class ActNormWrapper(EnvWrapperAct):
    def _process_act(self, act: np.ndarray) -> np.ndarray:
        lb, ub = self.wrapped_env.act_space.bounds
        return lb + (act + 1)*(ub - lb)/2

    def _process_act_space(self, space: BoxSpace) -> BoxSpace:
        ub = np.ones(space.shape)
        return BoxSpace(-ub, ub, labels=space.labels)
