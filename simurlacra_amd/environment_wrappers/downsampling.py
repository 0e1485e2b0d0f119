"""
DownsamplingWrapper (Pyrado/pyrado/environment_wrappers/downsampling.py:10-124), for policies that run at a lower rate
than the (500 Hz) device or simulation: of every `factor` consecutive actions only the first reaches the environment and is
repeated for the others; the observation handed out is a filter (the mean unless told otherwise) over the last `factor`
observations.

Host-only: the observation window is created once and survives reset() (as in the reference, :40-46,81-91), so its
state spans rollouts and is not local to a lane of the rollout kernels.  EnvPlan rejects the wrapper for fused launches;
rollout() then takes the step-by-step path, where every env.step() is one psh_step launch.
"""
from collections import deque
from functools import partial

import numpy as np

from .base import EnvWrapperAct, EnvWrapperObs


class DownsamplingWrapper(EnvWrapperAct, EnvWrapperObs):
    def __init__(self, wrapped_env, factor: int, obs_filter_fcn: callable = partial(np.mean, axis=0),
                 init_obs: np.ndarray = None):
        super().__init__(wrapped_env)
        if init_obs is not None and not isinstance(init_obs, np.ndarray):
            raise AssertionError('init_obs must be a NumPy array')
        self._factor = factor
        self._obs_filter_fcn = obs_filter_fcn
        # the window keeps the length it is given here, whatever `factor` is set to later
        self._window = deque([] if init_obs is None else [init_obs.copy()], maxlen=factor)
        self._restart()

    def _restart(self):
        self._step_in_rollout, self._held_act = 0, None

    @property
    def factor(self) -> int:
        return self._factor

    @factor.setter
    def factor(self, factor: int):
        assert isinstance(factor, int) and factor >= 1
        self._factor = factor
        self._step_in_rollout = 0

    def _save_domain_param(self, domain_param: dict):
        domain_param['downsampling'] = int(self._factor)

    def _load_domain_param(self, domain_param: dict):
        # (randomizers hand out 0-dim arrays or tensors)
        self._factor = int(domain_param.get('downsampling', self._factor))

    def reset(self, init_state: np.ndarray = None, domain_param: dict = None):
        if domain_param is not None:
            self._load_domain_param(domain_param)
        self._restart()
        return super().reset(init_state, domain_param)

    def _process_act(self, act: np.ndarray) -> np.ndarray:
        if self._factor < 1:
            return act
        fresh = self._step_in_rollout % self._factor == 0
        self._step_in_rollout += 1
        if fresh:
            self._held_act = act.copy()
            return act
        return self._held_act.copy()

    def _process_obs(self, obs: np.ndarray) -> np.ndarray:
        self._window.append(obs)
        return self._obs_filter_fcn(self._window)
