"""
Static definitions of the supported simulations: domain-parameter names in table-row order, nominal values
and default tasks.  Sources (relative to Pyrado/pyrado/environments/pysim/):
quanser_qube.py:38-44,61-71; quanser_cartpole.py:57-81,271-280,320-329; quanser_ball_balancer.py:80-152;
ball_on_beam.py:47-77; one_mass_oscillator.py:43-57; pendulum.py:30-47; quanser_qube.py:222-271.
"""
import numpy as np

from . import _abi as abi
from .specs import TaskSpec

KIND_BY_NAME = {'qq': abi.ENV_QQ, 'qcp-su': abi.ENV_QCP_SU, 'qcp-st': abi.ENV_QCP_ST, 'qbb': abi.ENV_QBB,
                'bob': abi.ENV_BOB, 'omo': abi.ENV_OMO, 'pend': abi.ENV_PEND, 'qq-st': abi.ENV_QQ_ST}

DP_NAMES = {
    abi.ENV_QQ: ('g', 'Rm', 'km', 'Mr', 'Lr', 'Dr', 'Mp', 'Lp', 'Dp', 'max_volt'),
    abi.ENV_QCP_SU: ('g', 'm_cart', 'l_rail', 'eta_m', 'eta_g', 'K_g', 'J_m', 'r_mp', 'R_m', 'k_m', 'B_pole', 'B_eq',
                     'm_pole', 'l_pole'),
    abi.ENV_QBB: ('g', 'm_ball', 'r_ball', 'l_plate', 'r_arm', 'K_g', 'eta_g', 'J_l', 'J_m', 'k_m', 'R_m', 'eta_m',
                  'B_eq', 'c_frict', 'V_thold_x_pos', 'V_thold_x_neg', 'V_thold_y_pos', 'V_thold_y_neg',
                  'offset_th_x', 'offset_th_y'),
    abi.ENV_BOB: ('g', 'm_ball', 'r_ball', 'm_beam', 'l_beam', 'd_beam', 'c_frict', 'ang_offset'),
    abi.ENV_OMO: ('m', 'k', 'd'),
    abi.ENV_PEND: ('g', 'm_pole', 'l_pole', 'd_pole', 'tau_max'),
}
DP_NAMES[abi.ENV_QCP_ST] = DP_NAMES[abi.ENV_QCP_SU]
DP_NAMES[abi.ENV_QQ_ST] = DP_NAMES[abi.ENV_QQ]

_NOMINAL = {
    abi.ENV_QQ: (9.81, 8.4, 0.042, 0.095, 0.085, 5e-6, 0.024, 0.129, 1e-6, 4.),
    abi.ENV_QBB: (9.81, 0.003, 0.019625, 0.275, 0.0254, 70., 0.9, 5.2822e-5, 4.6063e-7, 0.0077, 2.6, 0.69, 0.015,
                  0.05, 0.28, -0.10, 0.28, -0.074, 0., 0.),
    abi.ENV_BOB: (9.81, 0.5, 0.1, 3., 2., 0.1, 0.05, 0.),
    abi.ENV_OMO: (1., 30., 0.5),
    abi.ENV_PEND: (9.81, 1., 1., 0.05, 3.5),
}
_NOMINAL[abi.ENV_QQ_ST] = _NOMINAL[abi.ENV_QQ]


def nominal_domain_param(kind: int, long: bool = False) -> dict:
    """ Nominal domain parameters of a simulation as an ordered dict (table-row order). """
    if kind in (abi.ENV_QCP_SU, abi.ENV_QCP_ST):
        m_pole, l_pole = (0.23, 0.641/2) if long else (0.127, 0.3365/2)
        vals = (9.81, 0.38, 0.814, 0.9, 0.9, 3.71, 3.9e-7, 6.35e-3, 2.6, 7.67e-3, 0.0024, 5.4, m_pole, l_pole)
    else:
        vals = _NOMINAL[kind]
    return dict(zip(DP_NAMES[kind], vals))


def default_task_spec(kind: int, task_args: dict = None) -> TaskSpec:
    """ The task each simulation creates in _create_task(task_args). """
    ta = dict() if task_args is None else task_args
    if kind == abi.ENV_QQ:
        return TaskSpec(ta.get('state_des', np.array([0., np.pi, 0., 0.])), 'exp',
                        ta.get('Q', np.diag([3e-1, 1., 2e-2, 5e-3])), ta.get('R', np.diag([4e-3])), radial_idcs=[1])
    if kind == abi.ENV_QCP_SU:
        sd = ta.get('state_des', None)
        return TaskSpec(np.array([0., np.pi, 0., 0.]) if sd is None else sd, 'swingup', radial_idcs=[1],
                        final_mode=abi.FIN_ENABLED | abi.FIN_ALWAYS_NEGATIVE, final_factor=1e3)
    if kind == abi.ENV_QCP_ST:
        return TaskSpec(ta.get('state_des', np.array([0., np.pi, 0., 0.])), 'quadr',
                        ta.get('Q', np.diag([5e-0, 1e+1, 1e-2, 1e-2])), ta.get('R', np.diag([1e-3])), radial_idcs=[1],
                        final_mode=abi.FIN_ENABLED | abi.FIN_STATE_DEPENDENT | abi.FIN_TIME_DEPENDENT, final_factor=1e3)
    if kind == abi.ENV_QBB:
        return TaskSpec(ta.get('state_des', np.zeros(8)), 'scaledexp',
                        ta.get('Q', np.diag([1e0, 1e0, 5e3, 5e3, 1e-2, 1e-2, 5e-1, 5e-1])),
                        ta.get('R', np.diag([1e-2, 1e-2])), min_rew=1e-4)
    if kind == abi.ENV_BOB:
        return TaskSpec(ta.get('state_des', np.zeros(4)), 'scaledexp', ta.get('Q', np.diag([1e5, 1e3, 1e3, 1e2])),
                        ta.get('R', np.eye(1)), min_rew=1e-4)
    if kind == abi.ENV_OMO:
        return TaskSpec(ta.get('state_des', np.zeros(2)), 'quadr', ta.get('Q', np.diag([1e1, 1e-2])),
                        ta.get('R', np.diag([1e-6])), final_mode=abi.FIN_ENABLED | abi.FIN_ALWAYS_NEGATIVE,
                        final_factor=1e3)
    if kind == abi.ENV_PEND:  # the radial index is the velocity, as pendulum.py:36 has it
        return TaskSpec(ta.get('state_des', np.array([np.pi, 0.])), 'exp', ta.get('Q', np.diag([1e-0, 5e-3])),
                        ta.get('R', np.diag([1e-3])), radial_idcs=[1])
    if kind == abi.ENV_QQ_ST:  # quanser_qube.py:264-271: Q and R are fixed, only state_des comes from task_args
        sd = ta.get('state_des', None)
        return TaskSpec(np.array([0., np.pi, 0., 0.]) if sd is None else sd, 'exp', np.diag([3., 4., 2., 2.]),
                        np.diag([5e-2]), radial_idcs=[1])
    raise ValueError(f'unknown environment kind {kind}')
