"""
ObsNormWrapper (Pyrado/pyrado/environment_wrappers/observation_normalization.py:12-106): observations mapped
affinely from the wrapped environment's observation bounds (optionally overridden per label) to [-1, 1].
"""
from typing import Mapping, Optional

import numpy as np

from .. import inf
from ..spaces.box import BoxSpace
from ..utils.exceptions import ValueErr
from .base import EnvWrapperObs


class ObsNormWrapper(EnvWrapperObs):
    def __init__(self, wrapped_env, explicit_lb: Mapping[str, float] = None, explicit_ub: Mapping[str, float] = None):
        super().__init__(wrapped_env)
        self.explicit_lb = explicit_lb
        self.explicit_ub = explicit_ub

    @staticmethod
    def override_bounds(bounds: np.ndarray, override: Optional[Mapping[str, float]], bound_label: str,
                        names: np.ndarray) -> np.ndarray:
        if not override:
            return bounds
        bc = bounds.copy()
        for idx, name in np.ndenumerate(names):
            ov = override.get(name)
            if ov is not None:
                bc[idx] = ov
            elif np.isinf(bc[idx]):
                raise ValueErr(msg=f'{name} entry of {bound_label} bound is infinite and not overridden.'
                                   f'Cannot apply normalization.')
        return bc

    def _bounds(self, space):
        lb, ub = space.bounds
        lb = ObsNormWrapper.override_bounds(lb, self.explicit_lb, 'lower', space.labels)
        ub = ObsNormWrapper.override_bounds(ub, self.explicit_ub, 'upper', space.labels)
        return lb, ub

    def _process_obs(self, obs: np.ndarray) -> np.ndarray:
        lb, ub = self._bounds(self.wrapped_env.obs_space)
        return (obs - lb)/(ub - lb)*2 - 1

    def _process_obs_space(self, space: BoxSpace) -> BoxSpace:
        if not isinstance(space, BoxSpace):
            raise NotImplementedError('Only implemented ObsNormWrapper._process_obs_space() for BoxSpace!')
        lb, ub = space.bounds
        lb_ov, ub_ov = self._bounds(space)
        if any(lb_ov == -inf):
            raise ValueErr(msg=f'At least one element of the lower bounds is (negative) infinite:\n'
                               f'(overwritten) bound: {lb_ov}\nnames: {space.labels}')
        if any(ub_ov == inf):
            raise ValueErr(msg=f'At least one element of the upper bound is (positive) infinite:\n'
                               f'(overwritten) bound: {ub_ov}\nnames: {space.labels}')
        lb_norm = (lb - lb_ov)/(ub_ov - lb_ov)*2 - 1
        ub_norm = (ub - lb_ov)/(ub_ov - lb_ov)*2 - 1
        return BoxSpace(lb_norm, ub_norm, labels=space.labels)


class ObsRunningNormWrapper(EnvWrapperObs):
    """ Observations mapped to [-1, 1] with the bounds observed so far (observation_normalization.py:109-135;
    RunningNormalizer, Pyrado/pyrado/utils/normalizing.py:7-84).  Host-only: the normalizer's bounds grow over all
    rollouts of the object's life, so the wrapper is not local to a lane of the rollout kernels; EnvPlan rejects it and
    rollout() steps such an environment through psh_step. """

    def __init__(self, wrapped_env):
        super().__init__(wrapped_env)
        from ..utils.normalizing import RunningNormalizer
        self.normalizer = RunningNormalizer()

    def _process_obs(self, obs: np.ndarray) -> np.ndarray:
        return self.normalizer(obs)

    def _process_obs_space(self, space: BoxSpace) -> BoxSpace:
        if not isinstance(space, BoxSpace):
            raise NotImplementedError('Only implemented ObsRunningNormWrapper._process_obs_space() for BoxSpace!')
        one = np.ones(space.shape)
        return BoxSpace(-one, one, labels=space.labels)
