"""
ActNormWrapper (Pyrado/pyrado/environment_wrappers/action_normalization.py:7-22): actions in [-1, 1] are mapped
affinely onto the wrapped environment's current action bounds.
"""
import numpy as np

from ..spaces.box import BoxSpace
from .base import EnvWrapperAct


class ActNormWrapper(EnvWrapperAct):
    """ The policy sees the unit box; -1 and 1 are the bounds the wrapped environment has at the time of the step (they
    follow its domain parameters). """

    def _process_act_space(self, space: BoxSpace) -> BoxSpace:
        one = np.ones(space.shape)
        return BoxSpace(-one, one, labels=space.labels)

    def _process_act(self, act: np.ndarray) -> np.ndarray:
        lo, hi = self.wrapped_env.act_space.bounds
        # same order of operations as the reference (:13-15) and the fused kernels: lo + (a + 1) (hi - lo) / 2
        return lo + (act + 1)*(hi - lo)/2
