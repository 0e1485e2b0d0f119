"""
Default randomizers of the supported simulations and the BayRn index maps
(Pyrado/pyrado/domain_randomization/default_randomizers.py:41-97,128-263,382-429).
"""
import numpy as np

from .. import _abi as abi
from ..envdefs import nominal_domain_param
from .domain_parameter import NormalDomainParam, UniformDomainParam
from .domain_randomizer import DomainRandomizer


def _inner(env):
    while hasattr(env, 'wrapped_env'):
        env = env.wrapped_env
    return env


def _normal(nom, name, div, clip_lo):
    return NormalDomainParam(name=name, mean=nom[name], std=nom[name]/div, clip_lo=clip_lo)


def _uniform(nom, name, div, **kw):
    return UniformDomainParam(name=name, mean=nom[name], halfspan=nom[name]/div, **kw)


def get_default_randomizer_bob() -> DomainRandomizer:
    n = nominal_domain_param(abi.ENV_BOB)
    return DomainRandomizer(
        _normal(n, 'g', 5, 1e-4), _normal(n, 'm_ball', 5, 1e-4), _normal(n, 'r_ball', 5, 1e-4),
        _normal(n, 'm_beam', 5, 1e-3), _normal(n, 'l_beam', 5, 1e-3), _normal(n, 'd_beam', 5, 1e-3),
        UniformDomainParam(name='c_frict', mean=n['c_frict'], halfspan=n['c_frict'], clip_lo=0),
        UniformDomainParam(name='ang_offset', mean=0./180*np.pi, halfspan=0./180*np.pi))


def get_default_randomizer_omo() -> DomainRandomizer:
    n = nominal_domain_param(abi.ENV_OMO)
    return DomainRandomizer(_normal(n, 'm', 3, 1e-3), _normal(n, 'k', 3, 1e-3), _normal(n, 'd', 3, 1e-3))


def get_default_randomizer_pend() -> DomainRandomizer:
    n = nominal_domain_param(abi.ENV_PEND)
    return DomainRandomizer(_normal(n, 'g', 10, 1e-3), _normal(n, 'm_pole', 10, 1e-3), _normal(n, 'l_pole', 10, 1e-3),
                            _normal(n, 'd_pole', 10, 1e-3), _normal(n, 'tau_max', 10, 1e-3))


def get_default_randomizer_qbb() -> DomainRandomizer:
    n = nominal_domain_param(abi.ENV_QBB)
    return DomainRandomizer(
        # the reference names this first entry ']' (not 'g'), so gravity is never randomized; kept as is
        NormalDomainParam(name=']', mean=n['g'], std=n['g']/5, clip_lo=1e-4),
        _normal(n, 'm_ball', 5, 1e-4), _normal(n, 'r_ball', 5, 1e-3), _normal(n, 'l_plate', 5, 5e-2),
        _normal(n, 'r_arm', 5, 1e-4), _normal(n, 'K_g', 4, 1e-2), _normal(n, 'J_l', 4, 1e-6),
        _normal(n, 'J_m', 4, 1e-9), _normal(n, 'k_m', 4, 1e-4), _normal(n, 'R_m', 4, 1e-4),
        _uniform(n, 'eta_g', 4, clip_lo=1e-4, clip_up=1), _uniform(n, 'eta_m', 4, clip_lo=1e-4, clip_up=1),
        _uniform(n, 'B_eq', 4, clip_lo=1e-4), _uniform(n, 'c_frict', 4, clip_lo=1e-4),
        _uniform(n, 'V_thold_x_pos', 3), _uniform(n, 'V_thold_x_neg', 3),
        _uniform(n, 'V_thold_y_pos', 3), _uniform(n, 'V_thold_y_neg', 3),
        UniformDomainParam(name='offset_th_x', mean=n['offset_th_x'], halfspan=6./180*np.pi),
        UniformDomainParam(name='offset_th_y', mean=n['offset_th_y'], halfspan=6./180*np.pi))


def get_default_randomizer_qcp() -> DomainRandomizer:
    n = nominal_domain_param(abi.ENV_QCP_SU, long=False)
    return DomainRandomizer(
        _normal(n, 'g', 5, 1e-4), _normal(n, 'm_cart', 5, 1e-4), _normal(n, 'm_pole', 5, 1e-4),
        _normal(n, 'l_rail', 5, 1e-2), _normal(n, 'l_pole', 5, 1e-2),
        _uniform(n, 'eta_m', 4, clip_lo=1e-4, clip_up=1), _uniform(n, 'eta_g', 4, clip_lo=1e-4, clip_up=1),
        _normal(n, 'K_g', 4, 1e-4), _normal(n, 'J_m', 4, 1e-9), _normal(n, 'r_mp', 5, 1e-4),
        _normal(n, 'R_m', 4, 1e-4), _normal(n, 'k_m', 4, 1e-4),
        _uniform(n, 'B_eq', 4, clip_lo=1e-4), _uniform(n, 'B_pole', 4, clip_lo=1e-4))


def get_default_randomizer_qq() -> DomainRandomizer:
    n = nominal_domain_param(abi.ENV_QQ)
    return DomainRandomizer(
        _normal(n, 'g', 5, 1e-3), _normal(n, 'Rm', 5, 1e-3), _normal(n, 'km', 5, 1e-4), _normal(n, 'Mr', 5, 1e-4),
        _normal(n, 'Lr', 5, 1e-4), _normal(n, 'Dr', 5, 1e-9), _normal(n, 'Mp', 5, 1e-4), _normal(n, 'Lp', 5, 1e-4),
        _normal(n, 'Dp', 5, 1e-9))


def get_uniform_masses_lengths_randomizer_qq(frac_halfspan: float) -> DomainRandomizer:
    n = nominal_domain_param(abi.ENV_QQ)
    return DomainRandomizer(
        UniformDomainParam(name='Mp', mean=n['Mp'], halfspan=n['Mp']/frac_halfspan, clip_lo=1e-3),
        UniformDomainParam(name='Mr', mean=n['Mr'], halfspan=n['Mr']/frac_halfspan, clip_lo=1e-3),
        UniformDomainParam(name='Lr', mean=n['Lr'], halfspan=n['Lr']/frac_halfspan, clip_lo=1e-2),
        UniformDomainParam(name='Lp', mean=n['Lp'], halfspan=n['Lp']/frac_halfspan, clip_lo=1e-2))


_REGISTRY = {abi.ENV_BOB: get_default_randomizer_bob, abi.ENV_OMO: get_default_randomizer_omo,
             abi.ENV_QBB: get_default_randomizer_qbb, abi.ENV_QCP_SU: get_default_randomizer_qcp,
             abi.ENV_QCP_ST: get_default_randomizer_qcp, abi.ENV_QQ: get_default_randomizer_qq,
             abi.ENV_PEND: get_default_randomizer_pend,
             abi.ENV_QQ_ST: get_default_randomizer_qq}   # the reference finds QQubeSim's through the class hierarchy


def get_default_randomizer(env) -> DomainRandomizer:
    kind = getattr(_inner(env), 'kind', None)
    if kind not in _REGISTRY:
        raise ValueError(f'No default randomizer settings for env of type {type(_inner(env))}!')
    return _REGISTRY[kind]()


def get_conservative_randomizer(env) -> DomainRandomizer:
    r = get_default_randomizer(env)
    r.rescale_distr_param('std', 0.5)
    r.rescale_distr_param('cov', np.sqrt(0.5))
    r.rescale_distr_param('halfspan', 0.5)
    return r


def get_zero_var_randomizer(env) -> DomainRandomizer:
    r = get_default_randomizer(env)
    for p in ('std', 'cov', 'halfspan'):
        r.rescale_distr_param(p, 0.)
    return r


def get_empty_randomizer() -> DomainRandomizer:
    return DomainRandomizer()


def get_default_domain_param_map_omo():
    return {0: ('m', 'mean'), 1: ('m', 'std'), 2: ('k', 'mean'), 3: ('k', 'std'), 4: ('k', 'mean'), 5: ('d', 'std'),
            6: ('d', 'mean')}


def get_default_domain_param_map_qq():
    return {0: ('Mp', 'mean'), 1: ('Mp', 'std'), 2: ('Mr', 'mean'), 3: ('Mr', 'std'), 4: ('Lp', 'mean'),
            5: ('Lp', 'std'), 6: ('Lr', 'mean'), 7: ('Lr', 'std')}
