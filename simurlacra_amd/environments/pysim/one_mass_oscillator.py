"""
OneMassOscillatorSim.  Spaces, task, nominal parameters:
Pyrado/pyrado/environments/pysim/one_mass_oscillator.py:28-73.
"""
import numpy as np

from ... import _abi as abi
from ...envdefs import nominal_domain_param
from ...spaces import BoxSpace
from ...tasks import DesStateTask, FinalRewMode, FinalRewTask, QuadrErrRewFcn
from .base import SimPyEnv


class OneMassOscillatorSim(SimPyEnv):
    name: str = 'omo'
    kind = abi.ENV_OMO

    def _create_spaces(self):
        k = self._domain_param['k']
        max_state = np.array([1., 10.])
        min_init = np.array([-0.75*max_state[0], -0.01*max_state[1]])
        max_init = np.array([-0.65*max_state[0], +0.01*max_state[1]])
        max_act = np.array([max_state[0]*k])
        self._state_space = BoxSpace(-max_state, max_state, labels=['$x$', r'$\dot{x}$'])
        self._obs_space = self._state_space
        self._init_space = BoxSpace(min_init, max_init, labels=['$x$', r'$\dot{x}$'])
        self._act_space = BoxSpace(-max_act, max_act, labels=['$F$'])

    def _create_task(self, task_args: dict):
        state_des = task_args.get('state_des', np.zeros(2))
        Q = task_args.get('Q', np.diag([1e1, 1e-2]))
        R = task_args.get('R', np.diag([1e-6]))
        return FinalRewTask(DesStateTask(self.spec, state_des, QuadrErrRewFcn(Q, R)), factor=1e3,
                            mode=FinalRewMode(always_negative=True))

    @classmethod
    def get_nominal_domain_param(cls) -> dict:
        return nominal_domain_param(abi.ENV_OMO)

    def _calc_constants(self):
        m, k, d = (self._domain_param[n] for n in ('m', 'k', 'd'))
        self.omega = np.sqrt(k/m)
        self.zeta = d/(2.*np.sqrt(m*k))
