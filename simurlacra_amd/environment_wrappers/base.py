"""
EnvWrapper, EnvWrapperAct, EnvWrapperObs (Pyrado/pyrado/environment_wrappers/base.py:13-231): delegation of the
whole Env interface to the wrapped environment, with hooks to modify actions, observations and domain parameters.
"""
from abc import abstractmethod

import numpy as np

from ..environments.base import Env
from ..environments.sim_base import SimEnv
from ..utils.exceptions import TypeErr


def _forward(attr: str, writable: bool = False) -> property:
    """ A property that reads (and, if writable, writes) attribute `attr` of the wrapped environment. """
    def getter(self):
        return getattr(self._wrapped_env, attr)

    def setter(self, value):
        setattr(self._wrapped_env, attr, value)

    return property(getter, setter if writable else None, doc=f'`{attr}` of the wrapped environment')


class EnvWrapper(Env):
    """ Hands the whole Env interface down to the environment it wraps (:13-168); subclasses intercept what they change.
    Domain parameters pass two hooks on their way down and up, so that a wrapper can keep entries of its own in the dict. """

    def __init__(self, wrapped_env: Env):
        if not isinstance(wrapped_env, Env):
            raise TypeErr(given=wrapped_env, expected_type=Env)
        self._wrapped_env = wrapped_env

    wrapped_env = property(lambda self: self._wrapped_env)
    name = _forward('name')
    state = _forward('state', writable=True)
    state_space = _forward('state_space')
    obs_space = _forward('obs_space')
    act_space = _forward('act_space')
    dt = _forward('dt', writable=True)
    curr_step = _forward('curr_step')
    max_steps = _forward('max_steps', writable=True)
    task = _forward('task')
    supported_domain_param = _forward('supported_domain_param')

    @property
    def init_space(self):
        if not isinstance(self._wrapped_env, (SimEnv, EnvWrapper)):   # only simulations have one (:61-67)
            raise NotImplementedError
        return self._wrapped_env.init_space

    # domain parameters: hooks first on the way down, last on the way up ----------------------------------------------
    def _load_domain_param(self, param: dict):
        """ Called with the dict that is about to be set: take this wrapper's entries out of it. """

    def _save_domain_param(self, param: dict):
        """ Called with the dict that is about to be returned: put this wrapper's entries into it. """

    @property
    def domain_param(self) -> dict:
        collected = self._wrapped_env.domain_param
        self._save_domain_param(collected)
        return collected

    @domain_param.setter
    def domain_param(self, param: dict):
        self._load_domain_param(param)
        self._wrapped_env.domain_param = param

    def get_nominal_domain_param(self):
        return self._wrapped_env.get_nominal_domain_param()

    # stepping ---------------------------------------------------------------------------------------------------------
    def reset(self, init_state: np.ndarray = None, domain_param: dict = None) -> np.ndarray:
        if domain_param is not None:
            self._load_domain_param(domain_param)
        return self._wrapped_env.reset(init_state, domain_param)

    def step(self, act: np.ndarray) -> tuple:
        return self._wrapped_env.step(act)

    def observe(self, state):
        return self._wrapped_env.observe(state)

    def render(self, mode=None, render_step: int = 1):
        return self._wrapped_env.render(mode, render_step)

    def close(self):
        return self._wrapped_env.close()


class EnvWrapperAct(EnvWrapper):
    """ Changes the action on its way into step(), and the action space with it (:171-199). """

    @abstractmethod
    def _process_act(self, act: np.ndarray):
        raise NotImplementedError

    def _process_act_space(self, space):
        return space

    act_space = property(lambda self: self._process_act_space(super(EnvWrapperAct, self).act_space))

    def step(self, act: np.ndarray) -> tuple:
        return super().step(self._process_act(act))   # (cooperative: a wrapper may be both kinds, e.g. DownsamplingWrapper)


class EnvWrapperObs(EnvWrapper):
    """ Changes the observation on its way out of reset() and step(), and the observation space with it (:202-231). """

    @abstractmethod
    def _process_obs(self, obs: np.ndarray):
        raise NotImplementedError

    def _process_obs_space(self, space):
        return space

    obs_space = property(lambda self: self._process_obs_space(super(EnvWrapperObs, self).obs_space))

    def reset(self, init_state: np.ndarray = None, domain_param: dict = None) -> np.ndarray:
        return self._process_obs(super().reset(init_state, domain_param))

    def step(self, act: np.ndarray) -> tuple:
        out = super().step(act)
        return (self._process_obs(out[0]),) + tuple(out[1:])
