"""
DesStateTask and RadiallySymmDesStateTask (Pyrado/pyrado/tasks/desired_state.py:11-128).
"""
from typing import Callable, Sequence

import numpy as np

from ..utils.data_types import EnvSpec
from ..utils.exceptions import ShapeErr, TypeErr
from .base import Task, never_succeeded
from .reward_functions import RewFcn


class DesStateTask(Task):
    """ Reward on the error to a desired state (the desired action is zero) """

    def __init__(self, env_spec: EnvSpec, state_des: np.ndarray, rew_fcn: RewFcn, success_fcn: Callable = None):
        if not isinstance(env_spec, EnvSpec):
            raise TypeErr(given=env_spec, expected_type=EnvSpec)
        if not isinstance(state_des, np.ndarray):
            raise TypeErr(given=state_des, expected_type=np.ndarray)
        if not isinstance(rew_fcn, RewFcn):
            raise TypeErr(given=rew_fcn, expected_type=RewFcn)
        self._env_spec, self._state_des, self._rew_fcn = env_spec, state_des, rew_fcn
        self.success_fcn = success_fcn if success_fcn is not None else never_succeeded

    @property
    def env_spec(self):
        return self._env_spec

    @property
    def state_des(self):
        return self._state_des

    @state_des.setter
    def state_des(self, state_des):
        if not isinstance(state_des, np.ndarray):
            raise TypeErr(given=state_des, expected_type=np.ndarray)
        if state_des.shape != self._state_des.shape:
            raise ShapeErr(given=state_des, expected_match=self._state_des)
        self._state_des = state_des

    @property
    def rew_fcn(self):
        return self._rew_fcn

    @rew_fcn.setter
    def rew_fcn(self, rew_fcn):
        if not isinstance(rew_fcn, RewFcn):
            raise TypeErr(given=rew_fcn, expected_type=RewFcn)
        self._rew_fcn = rew_fcn

    def reset(self, env_spec: EnvSpec, state_des: np.ndarray = None, **kwargs):
        self._env_spec = env_spec
        if state_des is not None:
            self._state_des = state_des
        self._rew_fcn.reset(state_space=env_spec.state_space, act_space=env_spec.act_space, **kwargs)

    def step_rew(self, state, act, remaining_steps=None) -> float:
        return self._rew_fcn(self._state_des - state, -act, remaining_steps)

    def has_succeeded(self, state) -> bool:
        return self.success_fcn(self._state_des - state)


class RadiallySymmDesStateTask(DesStateTask):
    """ As DesStateTask with the errors of `idcs` taken modulo `modulation`; afterwards every error component beyond
    +-pi is folded back (all components, as the reference does). """

    def __init__(self, env_spec, state_des, rew_fcn, idcs: Sequence[int], modulation=2*np.pi, success_fcn=None):
        super().__init__(env_spec, state_des, rew_fcn, success_fcn)
        self.idcs = list(idcs)
        self.mod = modulation*np.ones(len(self.idcs))

    def step_rew(self, state, act, remaining_steps=None) -> float:
        err = self.state_des - state
        err[self.idcs] = np.fmod(err[self.idcs], self.mod)
        err[err > np.pi] = 2*np.pi - err[err > np.pi]
        err[err < -np.pi] = -2*np.pi - err[err < -np.pi]
        return self.rew_fcn(err, -act)
