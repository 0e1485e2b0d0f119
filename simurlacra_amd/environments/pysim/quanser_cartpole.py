"""
QCartPoleSim, QCartPoleStabSim, QCartPoleSwingUpSim.
Spaces, tasks and nominal parameters: Pyrado/pyrado/environments/pysim/quanser_cartpole.py:16-81,224-329.
"""
import numpy as np

from ... import _abi as abi, inf
from ...envdefs import nominal_domain_param
from ...spaces import BoxSpace
from ...tasks import (FinalRewMode, FinalRewTask, QuadrErrRewFcn, RadiallySymmDesStateTask,
                      UnderActuatedSwingUpRewFcn)
from .base import SimPyEnv


class QCartPoleSim(SimPyEnv):
    """ Base of the two cart-pole tasks """

    def __init__(self, dt: float, max_steps: int = inf, task_args: dict = None, long: bool = False):
        self._long = long
        self.x_buffer = 0.05  # [m] distance kept from the ends of the rail
        super().__init__(dt, max_steps, task_args)
        self.domain_param = self.get_nominal_domain_param(long=long)  # as the reference: re-set after the base init

    def _create_spaces(self):
        l_rail = self._domain_param['l_rail']
        max_obs = np.array([l_rail/2., 1., 1., np.inf, np.inf])
        self._obs_space = BoxSpace(-max_obs, max_obs, labels=['$x$', r'$\sin(\theta)$', r'$\cos(\theta)$', r'$\dot{x}$', r'$\dot{\theta}$'])
        self._act_space = BoxSpace(-np.array([8.]), np.array([8.]), labels=['$V$'])

    def observe(self, state):
        return np.array([state[0], np.sin(state[1]), np.cos(state[1]), state[2], state[3]])

    @classmethod
    def get_nominal_domain_param(cls, long: bool = False) -> dict:
        return nominal_domain_param(abi.ENV_QCP_SU, long=long)


class QCartPoleStabSim(QCartPoleSim):
    name: str = 'qcp-st'
    kind = abi.ENV_QCP_ST

    def __init__(self, dt: float, max_steps: int = inf, task_args: dict = None, long: bool = False):
        self.stab_thold = 15/180.*np.pi
        self.max_init_th_offset = 8/180.*np.pi
        super().__init__(dt, max_steps, task_args, long)

    def _create_spaces(self):
        super()._create_spaces()
        l_rail = self._domain_param['l_rail']
        min_state = np.array([-l_rail/2. + self.x_buffer, np.pi - self.stab_thold, -l_rail, -2*np.pi])
        max_state = np.array([+l_rail/2. - self.x_buffer, np.pi + self.stab_thold, +l_rail, +2*np.pi])
        max_init = np.array([+0.05, np.pi + self.max_init_th_offset, +0.05, +8/180*np.pi])
        min_init = np.array([-0.05, np.pi - self.max_init_th_offset, -0.05, -8/180*np.pi])
        labels = ['$x$', r'$\theta$', r'$\dot{x}$', r'$\dot{\theta}$']
        self._state_space = BoxSpace(min_state, max_state, labels=labels)
        self._init_space = BoxSpace(min_init, max_init, labels=labels)

    def _create_task(self, task_args: dict):
        state_des = task_args.get('state_des', np.array([0., np.pi, 0., 0.]))
        Q = task_args.get('Q', np.diag([5e-0, 1e+1, 1e-2, 1e-2]))
        R = task_args.get('R', np.diag([1e-3]))
        return FinalRewTask(RadiallySymmDesStateTask(self.spec, state_des, QuadrErrRewFcn(Q, R), idcs=[1]),
                            mode=FinalRewMode(state_dependent=True, time_dependent=True))


class QCartPoleSwingUpSim(QCartPoleSim):
    name: str = 'qcp-su'
    kind = abi.ENV_QCP_SU

    def _create_spaces(self):
        super()._create_spaces()
        l_rail = self._domain_param['l_rail']
        max_state = np.array([+l_rail/2. - self.x_buffer, +4*np.pi, np.inf, np.inf])
        min_state = np.array([-l_rail/2. + self.x_buffer, -4*np.pi, -np.inf, -np.inf])
        max_init = np.array([0.03, 1/180.*np.pi, 0.005, 2/180.*np.pi])
        labels = ['$x$', r'$\theta$', r'$\dot{x}$', r'$\dot{\theta}$']
        self._state_space = BoxSpace(min_state, max_state, labels=labels)
        self._init_space = BoxSpace(-max_init, max_init, labels=labels)

    def _create_task(self, task_args: dict):
        state_des = task_args.get('state_des', None)
        if state_des is None:
            state_des = np.array([0., np.pi, 0., 0.])
        return FinalRewTask(RadiallySymmDesStateTask(self.spec, state_des, UnderActuatedSwingUpRewFcn(), idcs=[1]),
                            mode=FinalRewMode(always_negative=True))
