"""
ActDelayWrapper (Pyrado/pyrado/environment_wrappers/action_delay.py:8-81): the action commanded `delay` steps ago is
applied; the queue starts with `delay` zero actions; the delay is the domain parameter 'act_delay'.
"""
import numpy as np

from .base import EnvWrapperAct


class ActDelayWrapper(EnvWrapperAct):
    def __init__(self, wrapped_env, delay: int = 0):
        super().__init__(wrapped_env)
        self._delay = delay
        self._act_queue = []

    @property
    def delay(self):
        return self._delay

    @delay.setter
    def delay(self, delay: int):
        assert isinstance(delay, int) and delay >= 0
        self._delay = delay

    def _save_domain_param(self, domain_param: dict):
        domain_param['act_delay'] = int(self._delay)

    def _load_domain_param(self, domain_param: dict):
        self._delay = int(domain_param.get('act_delay', self._delay))

    def reset(self, init_state: np.ndarray = None, domain_param: dict = None):
        if domain_param is not None:
            self._load_domain_param(domain_param)
        self._act_queue = [np.zeros(self.act_space.shape)]*self._delay
        return super().reset(init_state, domain_param)

    def _process_act(self, act: np.ndarray) -> np.ndarray:
        if self._delay != 0:
            self._act_queue.append(act)
            act = self._act_queue.pop(0)
        return act
