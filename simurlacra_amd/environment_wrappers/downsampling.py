"""
DownsamplingWrapper (Pyrado/pyrado/environment_wrappers/downsampling.py:10-124): every `factor`-th action is held for
`factor` steps, and the observation handed out is a filter (mean by default) over the last `factor` observations.

The observation buffer is not cleared by reset() in the reference (:81-91), so its state spans rollouts: the batched
rollout engine does not take this wrapper, it runs through the single-environment step loop of rollout().
"""
import functools
from collections import deque

import numpy as np

from .base import EnvWrapperAct, EnvWrapperObs


class DownsamplingWrapper(EnvWrapperAct, EnvWrapperObs):
    def __init__(self, wrapped_env, factor: int, obs_filter_fcn: callable = functools.partial(np.mean, axis=0),
                 init_obs: np.ndarray = None):
        super().__init__(wrapped_env)
        self._factor = factor
        self._cnt = 0
        self._act_last = None
        if init_obs is None:
            self._obs_buffer = deque([], maxlen=factor)
        else:
            assert isinstance(init_obs, np.ndarray)
            self._obs_buffer = deque([init_obs.copy()], maxlen=factor)
        self._obs_filter_fcn = obs_filter_fcn

    @property
    def factor(self):
        return self._factor

    @factor.setter
    def factor(self, factor: int):
        assert isinstance(factor, int) and factor >= 1
        self._factor = factor
        self._cnt = 0

    def _save_domain_param(self, domain_param: dict):
        domain_param['downsampling'] = int(self._factor)

    def _load_domain_param(self, domain_param: dict):
        self._factor = int(domain_param.get('downsampling', self._factor))

    def reset(self, init_state: np.ndarray = None, domain_param: dict = None):
        if domain_param is not None:
            self._load_domain_param(domain_param)
        self._act_last = None
        self._cnt = 0
        return super().reset(init_state, domain_param)

    def _process_act(self, act: np.ndarray) -> np.ndarray:
        if self._factor >= 1:
            if self._cnt % self._factor == 0:
                self._act_last = act.copy()
            else:
                act = self._act_last.copy()
            self._cnt += 1
        return act

    def _process_obs(self, obs: np.ndarray) -> np.ndarray:
        self._obs_buffer.append(obs)
        return self._obs_filter_fcn(self._obs_buffer)
