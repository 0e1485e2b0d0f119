"""
PendulumSim: under-actuated pendulum modelled as a rod.  Spaces, task, nominal parameters:
Pyrado/pyrado/environments/pysim/pendulum.py:12-47.
"""
import numpy as np

from ... import _abi as abi
from ...envdefs import nominal_domain_param
from ...spaces import BoxSpace
from ...spaces.singular import SingularStateSpace
from ...tasks import ExpQuadrErrRewFcn, RadiallySymmDesStateTask
from .base import SimPyEnv


class PendulumSim(SimPyEnv):
    name: str = 'pend'
    kind = abi.ENV_PEND

    def _create_spaces(self):
        tau_max = self._domain_param['tau_max']
        max_state = np.array([4*np.pi, 4*np.pi])
        max_obs = np.array([1., 1., np.inf])
        init_state = np.zeros(2)
        self._state_space = BoxSpace(-max_state, max_state, labels=[r'$\theta$', r'$\dot{\theta}$'])
        self._obs_space = BoxSpace(-max_obs, max_obs, labels=[r'$\sin\theta$', r'$\cos\theta$', r'$\dot{\theta}$'])
        self._init_space = SingularStateSpace(init_state, labels=[r'$\theta$', r'$\dot{\theta}$'])
        self._act_space = BoxSpace(-tau_max, tau_max, labels=[r'$\tau$'])

    def _create_task(self, task_args: dict):
        state_des = task_args.get('state_des', np.array([np.pi, 0.]))
        Q = task_args.get('Q', np.diag([1e-0, 5e-3]))
        R = task_args.get('R', np.diag([1e-3]))
        return RadiallySymmDesStateTask(self.spec, state_des, ExpQuadrErrRewFcn(Q, R), idcs=[1])

    def observe(self, state, dtype=np.ndarray):
        return np.array([np.sin(state[0]), np.cos(state[0]), state[1]])

    @classmethod
    def get_nominal_domain_param(cls) -> dict:
        return nominal_domain_param(abi.ENV_PEND)
