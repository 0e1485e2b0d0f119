"""
Reward functions of the supported simulations (Pyrado/pyrado/tasks/reward_functions.py:143-303).  The objects hold
the weights that the rollout kernels read; their __call__ evaluates one reward on the host for callers that use the
task objects directly.
"""
from abc import ABC, abstractmethod

import numpy as np

from ..spaces.base import Space
from ..utils.exceptions import TypeErr


class RewFcn(ABC):
    @abstractmethod
    def __call__(self, err_s: np.ndarray, err_a: np.ndarray, remaining_steps: int = None) -> float:
        raise NotImplementedError

    def reset(self, *args, **kwargs):
        pass


class QuadrErrRewFcn(RewFcn):
    """ r = -(e_s^T Q e_s + e_a^T R e_a) """
    kind = 'quadr'

    def __init__(self, Q, R):
        try:
            Q, R = np.array(Q, dtype=np.float64), np.array(R, dtype=np.float64)
        except Exception:
            raise TypeErr(given=Q, expected_type=[np.ndarray, list])
        assert (np.linalg.eigvals(Q).real >= 0).all(), 'Q must be positive semi-definite'
        assert (np.linalg.eigvals(R).real >= 0).all(), 'R must be positive semi-definite'
        self.Q, self.R = Q, R

    def _weighted_quadr_cost(self, err_s, err_a):
        return err_s.dot(self.Q.dot(err_s)) + err_a.dot(self.R.dot(err_a))

    def __call__(self, err_s, err_a, remaining_steps=None) -> float:
        assert isinstance(err_s, np.ndarray) and isinstance(err_a, np.ndarray)
        return -float(self._weighted_quadr_cost(err_s.reshape(-1), err_a.reshape(-1)))


class ExpQuadrErrRewFcn(QuadrErrRewFcn):
    """ r = exp(-cost) """
    kind = 'exp'

    def __call__(self, err_s, err_a, remaining_steps=None) -> float:
        return float(np.exp(-self._weighted_quadr_cost(err_s, err_a)))


class ScaledExpQuadrErrRewFcn(QuadrErrRewFcn):
    """ r = exp(-c_max*cost) with c_max = -ln(min_rew)/cost(worst-case state and action of the CURRENT spaces) """
    kind = 'scaledexp'

    def __init__(self, Q, R, state_space: Space, act_space: Space, min_rew: float = 1e-4):
        super().__init__(Q, R)
        self.c_max = None
        self.min_rew = min_rew
        self.reset(state_space, act_space, min_rew)

    def __call__(self, err_s, err_a, remaining_steps=None) -> float:
        return float(np.exp(-self.c_max*self._weighted_quadr_cost(err_s, err_a)))

    def reset(self, state_space: Space, act_space: Space, min_rew=None, **kwargs):
        if not isinstance(state_space, Space):
            raise TypeErr(given=state_space, expected_type=Space)
        if not isinstance(act_space, Space):
            raise TypeErr(given=act_space, expected_type=Space)
        if min_rew is not None:
            self.min_rew = min_rew
        obs_max = state_space.bound_abs_up.reshape(-1)
        act_max = act_space.bound_abs_up.reshape(-1)
        max_cost = obs_max.dot(self.Q.dot(obs_max)) + act_max.dot(self.R.dot(act_max))
        self.c_max = -1.0*np.log(self.min_rew)/max_cost


class UnderActuatedSwingUpRewFcn(RewFcn):
    """ Swing-up reward of the cart-pole: -(c_pole*(th^2 + ln(th^2 + 0.1)) + c_cart*|x| + c_act*u^2) + 10 """
    kind = 'swingup'

    def __init__(self, c_pole=1.0, c_cart=0.2, c_act=1e-3, c_theta_sq=1.0, c_theta_log=0.1, idx_x=0, idx_th=1):
        self.c_costs = np.array([c_pole, c_cart, c_act])
        self.c_theta_sq, self.c_theta_log = c_theta_sq, c_theta_log
        self.idx_x, self.idx_th = idx_x, idx_th

    def __call__(self, err_s, err_a, remaining_steps=None) -> float:
        eth = err_s[self.idx_th]
        cost_pole = self.c_theta_sq*eth**2 + np.log(eth**2 + self.c_theta_log)
        cost_cart = np.abs(err_s[self.idx_x])
        cost_act = float(np.asarray(err_a).reshape(-1)[0])**2
        return float(-self.c_costs @ np.asarray([cost_pole, cost_cart, cost_act]) + 10)
