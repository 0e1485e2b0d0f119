"""
Env: the environment interface of Pyrado (Pyrado/pyrado/environments/base.py:13-190).
"""
from abc import ABC, abstractmethod

import numpy as np

from .. import inf
from ..utils.data_types import EnvSpec, RenderMode
from ..utils.exceptions import TypeErr, ValueErr


class Env(ABC):
    name: str = None

    def __init__(self, dt, max_steps=inf):
        if dt is not None and dt < 0:
            raise ValueErr(given=dt, eq_constraint='None', ge_constraint='0')
        if max_steps < 1:
            raise ValueErr(given=max_steps, ge_constraint='1')
        self._dt, self._max_steps = dt, max_steps
        self._curr_step = 0
        self._curr_rew = -inf
        self.state = None

    @property
    @abstractmethod
    def state_space(self):
        raise NotImplementedError

    @property
    @abstractmethod
    def obs_space(self):
        raise NotImplementedError

    @property
    @abstractmethod
    def act_space(self):
        raise NotImplementedError

    @property
    def spec(self) -> EnvSpec:
        return EnvSpec(self.obs_space, self.act_space, self.state_space)

    @property
    def dt(self):
        return self._dt

    @dt.setter
    def dt(self, dt):
        if not dt > 0:
            raise ValueErr(given=dt, g_constraint='0')
        if not isinstance(dt, (float, int)):
            raise TypeErr(given=dt, expected_type=[float, int])
        self._dt = float(dt)

    @property
    def max_steps(self):
        return self._max_steps

    @max_steps.setter
    def max_steps(self, num_steps):
        if not (isinstance(num_steps, (int, np.integer)) or num_steps == inf):
            raise TypeErr(msg=f'Number of steps needs to be an integer of infinite, but is {num_steps}')
        if not num_steps > 0:
            raise ValueErr(given=num_steps, g_constraint='0')
        self._max_steps = num_steps

    @property
    def curr_step(self) -> int:
        return self._curr_step

    @property
    @abstractmethod
    def task(self):
        raise NotImplementedError

    @abstractmethod
    def reset(self, init_state: np.ndarray = None, domain_param: dict = None) -> np.ndarray:
        raise NotImplementedError

    @abstractmethod
    def step(self, act: np.ndarray) -> tuple:
        raise NotImplementedError

    def observe(self, state: np.ndarray) -> np.ndarray:
        return state.copy()

    def limit_act(self, act: np.ndarray) -> np.ndarray:
        return self.act_space.project_to(act)

    def render(self, mode: RenderMode = RenderMode(), render_step: int = 1):
        pass

    def close(self):
        pass
