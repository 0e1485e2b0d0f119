"""
Env: the environment interface of Pyrado (Pyrado/pyrado/environments/base.py:13-190).
"""
from abc import ABC, abstractmethod

import numpy as np

from .. import inf
from ..utils.data_types import EnvSpec, RenderMode
from ..utils.exceptions import TypeErr, ValueErr


def required(name: str, doc: str) -> property:
    """ A read-only property that every concrete class has to provide. """
    def getter(self):
        raise NotImplementedError(f'{type(self).__name__}.{name}')
    getter.__name__ = name
    return property(abstractmethod(getter), doc=doc)


class Env(ABC):
    """ What rollout() and the samplers need of an environment: three spaces, a task, a time step, a step limit and the
    reset / step pair.  Time step and step limit are validated where they are set (:36-47, :103-137). """
    name: str = None

    state_space = required('state_space', 'space of the states')
    obs_space = required('obs_space', 'space of the observations')
    act_space = required('act_space', 'space of the actions')
    task = required('task', 'the task that yields the step reward and decides about done')

    def __init__(self, dt, max_steps=inf):
        if dt is not None and dt < 0:
            raise ValueErr(given=dt, eq_constraint='None', ge_constraint='0')
        if max_steps < 1:
            raise ValueErr(given=max_steps, ge_constraint='1')
        self._dt, self._max_steps = dt, max_steps
        self._curr_step, self._curr_rew = 0, -inf
        self.state = None

    # time step and step limit ---------------------------------------------------------------------------------------------
    def _get_dt(self):
        return self._dt

    def _set_dt(self, value):
        if not value > 0:
            raise ValueErr(given=value, g_constraint='0')
        if not isinstance(value, (float, int)):
            raise TypeErr(given=value, expected_type=[float, int])
        self._dt = float(value)

    def _get_max_steps(self):
        return self._max_steps

    def _set_max_steps(self, value):
        whole = isinstance(value, (int, np.integer))
        if not whole and value != inf:
            raise TypeErr(msg=f'Number of steps needs to be an integer of infinite, but is {value}')
        if not value > 0:
            raise ValueErr(given=value, g_constraint='0')
        self._max_steps = value

    dt = property(_get_dt, _set_dt)
    max_steps = property(_get_max_steps, _set_max_steps)
    curr_step = property(lambda self: self._curr_step)
    spec = property(lambda self: EnvSpec(self.obs_space, self.act_space, self.state_space))

    # stepping -------------------------------------------------------------------------------------------------------------
    @abstractmethod
    def reset(self, init_state: np.ndarray = None, domain_param: dict = None) -> np.ndarray:
        raise NotImplementedError

    @abstractmethod
    def step(self, act: np.ndarray) -> tuple:
        raise NotImplementedError

    def observe(self, state: np.ndarray) -> np.ndarray:
        """ Fully observable unless a simulation says otherwise. """
        return state.copy()

    def limit_act(self, act: np.ndarray) -> np.ndarray:
        return self.act_space.project_to(act)

    def render(self, mode: RenderMode = RenderMode(), render_step: int = 1):
        """ Nothing to show by default. """

    def close(self):
        """ Nothing to release by default. """
