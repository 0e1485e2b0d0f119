"""
ObsPartialWrapper (Pyrado/pyrado/environment_wrappers/observation_partial.py:10-47): drops (or keeps only) selected
entries of the observation.
"""
import numpy as np

from ..utils.exceptions import ShapeErr
from .base import EnvWrapperObs


class ObsPartialWrapper(EnvWrapperObs):
    def __init__(self, wrapped_env, mask: list = None, idcs: list = None, keep_selected: bool = False):
        super().__init__(wrapped_env)
        if mask is not None:
            mask = np.array(mask, dtype=bool)
            if not mask.shape == wrapped_env.obs_space.shape:
                raise ShapeErr(given=mask, expected_match=wrapped_env.obs_space)
        else:
            assert idcs is not None, 'Either mask or indices must be specified'
            mask = wrapped_env.obs_space.create_mask(idcs)
        self.keep_mask = mask if keep_selected else np.logical_not(mask)

    def _process_obs(self, obs: np.ndarray) -> np.ndarray:
        return obs[self.keep_mask]

    def _process_obs_space(self, space):
        return space.subspace(self.keep_mask)
