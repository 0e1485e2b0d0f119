"""
EnvWrapper, EnvWrapperAct, EnvWrapperObs (Pyrado/pyrado/environment_wrappers/base.py:13-231): delegation of the
whole Env interface to the wrapped environment, with hooks to modify actions, observations and domain parameters.
"""
from abc import abstractmethod

import numpy as np

from ..environments.base import Env
from ..environments.sim_base import SimEnv
from ..utils.exceptions import TypeErr


class EnvWrapper(Env):
    def __init__(self, wrapped_env: Env):
        if not isinstance(wrapped_env, Env):
            raise TypeErr(given=wrapped_env, expected_type=Env)
        self._wrapped_env = wrapped_env

    @property
    def name(self) -> str:
        return self._wrapped_env.name

    @property
    def wrapped_env(self):
        return self._wrapped_env

    @property
    def state(self):
        return self._wrapped_env.state

    @state.setter
    def state(self, value):
        self._wrapped_env.state = value

    @property
    def state_space(self):
        return self._wrapped_env.state_space

    @property
    def obs_space(self):
        return self._wrapped_env.obs_space

    @property
    def act_space(self):
        return self._wrapped_env.act_space

    @property
    def init_space(self):
        if isinstance(self._wrapped_env, (SimEnv, EnvWrapper)):
            return self._wrapped_env.init_space
        raise NotImplementedError

    @property
    def dt(self):
        return self._wrapped_env.dt

    @dt.setter
    def dt(self, dt):
        self._wrapped_env.dt = dt

    @property
    def curr_step(self) -> int:
        return self._wrapped_env.curr_step

    @property
    def max_steps(self):
        return self._wrapped_env.max_steps

    @max_steps.setter
    def max_steps(self, num_steps):
        self._wrapped_env.max_steps = num_steps

    @property
    def task(self):
        return self._wrapped_env.task

    @property
    def domain_param(self) -> dict:
        param = self._wrapped_env.domain_param
        self._save_domain_param(param)
        return param

    @domain_param.setter
    def domain_param(self, param: dict):
        self._load_domain_param(param)
        self._wrapped_env.domain_param = param

    @property
    def supported_domain_param(self):
        return self._wrapped_env.supported_domain_param

    def get_nominal_domain_param(self):
        return self._wrapped_env.get_nominal_domain_param()

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

    def _load_domain_param(self, param: dict):
        pass

    def _save_domain_param(self, param: dict):
        pass


class EnvWrapperAct(EnvWrapper):
    @abstractmethod
    def _process_act(self, act: np.ndarray):
        raise NotImplementedError

    def _process_act_space(self, space):
        return space

    def step(self, act: np.ndarray) -> tuple:
        return super().step(self._process_act(act))

    @property
    def act_space(self):
        return self._process_act_space(super().act_space)


class EnvWrapperObs(EnvWrapper):
    @abstractmethod
    def _process_obs(self, obs: np.ndarray):
        raise NotImplementedError

    def _process_obs_space(self, space):
        return space

    @property
    def obs_space(self):
        return self._process_obs_space(super().obs_space)

    def reset(self, init_state: np.ndarray = None, domain_param: dict = None) -> np.ndarray:
        return self._process_obs(super().reset(init_state, domain_param))

    def step(self, act: np.ndarray) -> tuple:
        obs, rew, done, info = super().step(act)
        return self._process_obs(obs), rew, done, info
