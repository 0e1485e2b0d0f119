"""
QBallBalancerSim and its inverse kinematics QBallBalancerKin.
Spaces, task, nominal parameters: Pyrado/pyrado/environments/pysim/quanser_ball_balancer.py:34-195; IK :380-480.
The measured voltage thresholds of the reference are looked up in a directory that its data does not ship in, so the
hard-coded defaults (:99-102) are the nominal values.
"""
import numpy as np

from ... import _abi as abi, inf
from ...envdefs import nominal_domain_param
from ...spaces import BoxSpace, Polar2DPosVelSpace
from ...tasks import DesStateTask, ScaledExpQuadrErrRewFcn
from .base import SimPyEnv


class QBallBalancerSim(SimPyEnv):
    name: str = 'qbb'
    kind = abi.ENV_QBB

    def __init__(self, dt: float, max_steps: int = inf, task_args: dict = None, simplified_dyn: bool = False,
                 load_experimental_tholds: bool = True):
        self._simplified_dyn = simplified_dyn
        super().__init__(dt, max_steps, task_args)
        if not simplified_dyn:
            self._kin = QBallBalancerKin(self)

    @property
    def simplified_dyn(self) -> bool:
        return self._simplified_dyn

    def _num_hidden(self) -> int:
        return 2

    @property
    def plate_angs(self) -> np.ndarray:
        """ Plate angles alpha, beta: integrator state set by the inverse kinematics at reset """
        return self._hidden

    @plate_angs.setter
    def plate_angs(self, value):
        self._hidden = np.asarray(value, dtype=np.float64).reshape(2).copy()

    def _create_spaces(self):
        l_plate = self._domain_param['l_plate']
        max_state = np.array([np.pi/4., np.pi/4., l_plate/2., l_plate/2., 5*np.pi, 5*np.pi, 0.5, 0.5])
        min_init = np.array([0.75*l_plate/2, -np.pi, -0.05*max_state[6], -0.05*max_state[7]])
        max_init = np.array([0.8*l_plate/2, np.pi, 0.05*max_state[6], 0.05*max_state[7]])
        max_act = np.array([3.0, 3.0])
        self._state_space = BoxSpace(-max_state, max_state, labels=[r'$\theta_x$', r'$\theta_y$', '$x$', '$y$', r'$\dot{\theta}_x$',
                                                                   r'$\dot{\theta}_y$', r'$\dot{x}$', r'$\dot{y}$'])
        self._obs_space = self._state_space.copy()
        self._init_space = Polar2DPosVelSpace(min_init, max_init, labels=['$r$', r'$\phi$', r'$\dot{x}$', r'$\dot{y}$'])
        self._act_space = BoxSpace(-max_act, max_act, labels=['$V_x$', '$V_y$'])

    def _create_task(self, task_args: dict):
        state_des = task_args.get('state_des', np.zeros(8))
        Q = task_args.get('Q', np.diag([1e0, 1e0, 5e3, 5e3, 1e-2, 1e-2, 5e-1, 5e-1]))
        R = task_args.get('R', np.diag([1e-2, 1e-2]))
        return DesStateTask(self.spec, state_des,
                            ScaledExpQuadrErrRewFcn(Q, R, self.state_space, self.act_space, min_rew=1e-4))

    @classmethod
    def get_V_tholds(cls, load_experiments: bool = True) -> dict:
        return dict(V_thold_x_pos=0.28, V_thold_x_neg=-0.10, V_thold_y_pos=0.28, V_thold_y_neg=-0.074)

    @classmethod
    def get_nominal_domain_param(cls) -> dict:
        return nominal_domain_param(abi.ENV_QBB)

    def _state_from_init(self, init_state):
        state = np.zeros(8)
        state[2:4] = init_state[:2]
        state[6:8] = init_state[2:]
        return state


class QBallBalancerKin:
    """ Plate angle from the servo shaft angle for one axis (100 momentum-SGD steps on the two-link constraint loss,
    evaluated on the device by psh_qbb_ik). """

    def __init__(self, qbb, num_opt_iter: int = 100):
        if num_opt_iter != 100:
            raise NotImplementedError('the device IK runs the reference default of 100 iterations')
        self._qbb = qbb
        self.num_opt_iter = num_opt_iter
        self.d = 0.10

    def __call__(self, th) -> float:
        from ...engine import get_engine
        self.r = float(self._qbb.domain_param['r_arm'])
        self.l = float(self._qbb.domain_param['l_plate']/2.)
        ang = get_engine().qbb_ik(np.array([float(th)]), np.array([self.r]), np.array([2.*self.l]))
        return float(ang.item())
