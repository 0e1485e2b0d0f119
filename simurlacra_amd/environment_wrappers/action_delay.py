"""
ActDelayWrapper (Pyrado/pyrado/environment_wrappers/action_delay.py:8-81): the wrapped environment receives the action
commanded `delay` steps earlier; before that, zeros.  The delay is the integer domain parameter 'act_delay', so a
randomizer can draw it per rollout.

The pipeline holds the caller's array OBJECTS, as the reference's list does: an environment that writes into the action
it receives (QBallBalancerSim's dead zone) thereby changes an action the caller recorded `delay` steps ago.  The fused
kernels reproduce that (psh_wrappers, ACT_DELAY).
"""
from collections import deque

import numpy as np

from .base import EnvWrapperAct


class ActDelayWrapper(EnvWrapperAct):
    def __init__(self, wrapped_env, delay: int = 0):
        super().__init__(wrapped_env)
        self._pipeline = deque()
        self.delay = delay

    @property
    def delay(self) -> int:
        return self._delay

    @delay.setter
    def delay(self, steps: int):
        if not isinstance(steps, (int, np.integer)) or steps < 0:
            raise AssertionError(f'the action delay must be a non-negative integer, got {steps!r}')
        self._delay = int(steps)

    def _save_domain_param(self, domain_param: dict):
        domain_param['act_delay'] = self._delay

    def _load_domain_param(self, domain_param: dict):
        if 'act_delay' in domain_param:
            self._delay = int(domain_param['act_delay'])

    def reset(self, init_state: np.ndarray = None, domain_param: dict = None):
        if domain_param is not None:
            self._load_domain_param(domain_param)
        idle = np.zeros(self.act_space.shape)
        self._pipeline = deque(idle for _ in range(self._delay))   # one shared zero action, as in the reference
        return super().reset(init_state, domain_param)

    def _process_act(self, act: np.ndarray) -> np.ndarray:
        if self._delay == 0:
            return act
        self._pipeline.append(act)
        return self._pipeline.popleft()
