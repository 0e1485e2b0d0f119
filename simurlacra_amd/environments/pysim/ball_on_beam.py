"""
BallOnBeamSim.  Spaces, task, nominal parameters: Pyrado/pyrado/environments/pysim/ball_on_beam.py:21-77.
"""
import numpy as np

from ... import _abi as abi
from ...envdefs import nominal_domain_param
from ...spaces import BoxSpace, CompoundSpace
from ...tasks import DesStateTask, ScaledExpQuadrErrRewFcn
from .base import SimPyEnv


class BallOnBeamSim(SimPyEnv):
    name: str = 'bob'
    kind = abi.ENV_BOB

    def _create_spaces(self):
        l_beam, g = self._domain_param['l_beam'], self._domain_param['g']
        max_state = np.array([l_beam/2., np.pi/4., 10., np.pi])
        max_act = np.array([l_beam/2.*g*3.])
        labels = ['$x$', r'$\alpha$', r'$\dot{x}$', r'$\dot{\alpha}$']
        self._state_space = BoxSpace(-max_state, max_state, labels=labels)
        self._obs_space = self._state_space
        self._init_space = CompoundSpace([
            BoxSpace(np.array([-0.8*l_beam/2., -5/180.*np.pi, -0.02*max_state[2], -0.02*max_state[3]]),
                     np.array([-0.7*l_beam/2., +5/180.*np.pi, +0.02*max_state[2], +0.02*max_state[3]]), labels=labels),
            BoxSpace(np.array([0.7*l_beam/2., -5/180.*np.pi, -0.02*max_state[2], -0.02*max_state[3]]),
                     np.array([0.8*l_beam/2., +5/180.*np.pi, +0.02*max_state[2], +0.02*max_state[3]]), labels=labels),
        ])
        self._act_space = BoxSpace(-max_act, max_act, labels=[r'$\tau$'])

    def _create_task(self, task_args: dict):
        state_des = task_args.get('state_des', np.zeros(4))
        Q = task_args.get('Q', np.diag([1e5, 1e3, 1e3, 1e2]))
        R = task_args.get('R', np.eye(1))
        return DesStateTask(self.spec, state_des,
                            ScaledExpQuadrErrRewFcn(Q, R, self.state_space, self.act_space, min_rew=1e-4))

    @classmethod
    def get_nominal_domain_param(cls) -> dict:
        return nominal_domain_param(abi.ENV_BOB)
