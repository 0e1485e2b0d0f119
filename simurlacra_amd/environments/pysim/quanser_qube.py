"""
QQubeSim (Furuta pendulum swing-up; later upstream versions call it QQubeSwingUpSim).
Spaces, task and nominal parameters: Pyrado/pyrado/environments/pysim/quanser_qube.py:20-71.
"""
import numpy as np

from ... import _abi as abi
from ...envdefs import nominal_domain_param
from ...spaces import BoxSpace
from ...tasks import ExpQuadrErrRewFcn, RadiallySymmDesStateTask
from .base import SimPyEnv


class QQubeSim(SimPyEnv):
    name: str = 'qq'
    kind = abi.ENV_QQ

    def _create_spaces(self):
        max_volt = self._domain_param['max_volt']
        max_state = np.array([115./180*np.pi, 4*np.pi, 20*np.pi, 20*np.pi])
        max_obs = np.array([1., 1., 1., 1., np.inf, np.inf])
        max_init_state = np.array([2., 1., 0.5, 0.5])/180*np.pi
        self._state_space = BoxSpace(-max_state, max_state, labels=[r'$\theta$', r'$\alpha$', r'$\dot{\theta}$', r'$\dot{\alpha}$'])
        self._obs_space = BoxSpace(-max_obs, max_obs, labels=[r'$\sin\theta$', r'$\cos\theta$', r'$\sin\alpha$', r'$\cos\alpha$',
                                                              r'$\dot{\theta}$', r'$\dot{\alpha}$'])
        self._init_space = BoxSpace(-max_init_state, max_init_state, labels=[r'$\theta$', r'$\alpha$', r'$\dot{\theta}$', r'$\dot{\alpha}$'])
        self._act_space = BoxSpace(-max_volt, max_volt, labels=['$V$'])

    def _create_task(self, task_args: dict):
        state_des = task_args.get('state_des', np.array([0., np.pi, 0., 0.]))
        Q = task_args.get('Q', np.diag([3e-1, 1., 2e-2, 5e-3]))
        R = task_args.get('R', np.diag([4e-3]))
        return RadiallySymmDesStateTask(self.spec, state_des, ExpQuadrErrRewFcn(Q, R), idcs=[1])

    def observe(self, state, dtype=np.ndarray):
        return np.array([np.sin(state[0]), np.cos(state[0]), np.sin(state[1]), np.cos(state[1]), state[2], state[3]])

    @classmethod
    def get_nominal_domain_param(cls) -> dict:
        return nominal_domain_param(abi.ENV_QQ)


QQubeSwingUpSim = QQubeSim


class QQubeStabSim(QQubeSim):
    """ Stabilisation of the pendulum around the upright position (Pyrado/pyrado/environments/pysim/quanser_qube.py:222-271):
    the dynamics of QQubeSim, a state space that ends 15 deg from upright, the error to upright as observation. """
    name: str = 'qq-st'
    kind = abi.ENV_QQ_ST

    def _create_spaces(self):
        max_volt = self._domain_param['max_volt']
        self.stab_thold = 15/180.*np.pi
        min_state = np.array([-120./180*np.pi, np.pi - self.stab_thold, -20*np.pi, -20*np.pi])
        max_state = np.array([120./180*np.pi, np.pi + self.stab_thold, 20*np.pi, 20*np.pi])
        min_init_state = np.array([-5./180*np.pi, 175./180*np.pi, 0, 0])
        max_init_state = np.array([5./180*np.pi, 185./180*np.pi, 0, 0])
        labels = [r'$\theta$', r'$\alpha$', r'$\dot{\theta}$', r'$\dot{\alpha}$']
        self._state_space = BoxSpace(min_state, max_state, labels=labels)
        self._obs_space = self._state_space
        self._init_space = BoxSpace(min_init_state, max_init_state, labels=labels)
        self._act_space = BoxSpace(-max_volt, max_volt, labels=['$V$'])

    def observe(self, state, dtype=np.ndarray):
        return (np.array([0., np.pi, 0., 0.]) - state).copy()

    def _create_task(self, task_args: dict):
        state_des = task_args.get('state_des', None)
        if state_des is None:
            state_des = np.array([0., np.pi, 0., 0.])
        Q = np.diag([3., 4., 2., 2.])
        R = np.diag([5e-2])
        return RadiallySymmDesStateTask(self.spec, state_des, ExpQuadrErrRewFcn(Q, R), idcs=[1])
