"""
Seeded synthetic inputs shared by the CPU and GPU tests: per-environment initial states drawn from the
environment's own init space, domain parameters perturbed around the nominal values, and random policies.
"""
import numpy as np

from simurlacra_amd import _abi as abi, envdefs
from simurlacra_amd.specs import WrapSpec, PolicySpec

KINDS = [abi.ENV_QQ, abi.ENV_QCP_SU, abi.ENV_QCP_ST, abi.ENV_QBB, abi.ENV_BOB, abi.ENV_OMO, abi.ENV_PEND, abi.ENV_QQ_ST]
KIND_NAMES = {abi.ENV_QQ: 'qq', abi.ENV_QCP_SU: 'qcp-su', abi.ENV_QCP_ST: 'qcp-st', abi.ENV_QBB: 'qbb',
              abi.ENV_BOB: 'bob', abi.ENV_OMO: 'omo', abi.ENV_PEND: 'pend', abi.ENV_QQ_ST: 'qq-st'}
DT = {abi.ENV_QQ: 0.004, abi.ENV_QCP_SU: 0.002, abi.ENV_QCP_ST: 0.002, abi.ENV_QBB: 0.01, abi.ENV_BOB: 0.01,
      abi.ENV_OMO: 0.01, abi.ENV_PEND: 0.02, abi.ENV_QQ_ST: 0.004}


def domain_params(kind, n, rng, spread=0.1):
    """ [P, n] table: nominal values with a relative Gaussian perturbation (signs preserved). """
    nom = np.array(list(envdefs.nominal_domain_param(kind).values()))
    dp = nom[:, None]*(1 + spread*rng.standard_normal((nom.size, n)))
    if kind == abi.ENV_QBB:
        dp[18:20] = 0.05*rng.standard_normal((2, n))   # servo offsets (nominal 0)
    if kind == abi.ENV_BOB:
        dp[7] = 0.01*rng.standard_normal(n)            # beam angle offset (nominal 0)
    return np.ascontiguousarray(dp)


def init_states(kind, dp, rng):
    """ [init_dim, n] initial states inside each environment's init space. """
    n = dp.shape[1]
    pi = np.pi
    if kind == abi.ENV_QQ:
        hw = np.array([2, 1, .5, .5])/180*pi
        return rng.uniform(-1, 1, (4, n))*hw[:, None]
    if kind == abi.ENV_QCP_SU:
        hw = np.array([.03, pi/180, .005, 2*pi/180])
        return rng.uniform(-1, 1, (4, n))*hw[:, None]
    if kind == abi.ENV_PEND:   # the init space is the single zero state; the tests also start elsewhere
        return np.stack([rng.uniform(-3., 3., n), rng.uniform(-1., 1., n)])
    if kind == abi.ENV_QQ_ST:
        return np.stack([rng.uniform(-5, 5, n)/180*pi, rng.uniform(175, 185, n)/180*pi, np.zeros(n), np.zeros(n)])
    if kind == abi.ENV_QCP_ST:
        return np.stack([rng.uniform(-.05, .05, n), pi + rng.uniform(-.13, .13, n), rng.uniform(-.05, .05, n),
                         rng.uniform(-.13, .13, n)])
    if kind == abi.ENV_QBB:
        r = rng.uniform(.75, .8, n)*dp[3]/2
        ph = rng.uniform(-pi, pi, n)
        return np.stack([r*np.cos(ph), r*np.sin(ph), rng.uniform(-.025, .025, n), rng.uniform(-.025, .025, n)])
    if kind == abi.ENV_BOB:
        return np.stack([rng.choice([-1, 1], n)*rng.uniform(.7, .8, n)*dp[4]/2, rng.uniform(-.08, .08, n),
                         rng.uniform(-.2, .2, n), rng.uniform(-.06, .06, n)])
    return np.stack([rng.uniform(-.75, -.65, n), rng.uniform(-.1, .1, n)])


def linear_policy(kind, rng, feats=('const', 'identity', 'squared'), scale=0.5, obs_dim=None):
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    ps = PolicySpec('linear', O if obs_dim is None else obs_dim, A, feats=list(feats))
    return ps, scale*rng.standard_normal(ps.num_param)


def structured_feats(O, rng, state_wise=True, D=24, K=5):
    """ A FeatureStack with MultFeat, RandFourierFeat and RBFFeat as FeatSpecs (data as their constructors compute it,
    Pyrado/pyrado/policies/features.py:146-154,192-201,262-276). """
    from simurlacra_amd.specs import FeatSpec
    freq = rng.standard_normal((D, O))*np.sqrt(2./np.full((1, O), 2.0))
    centers = np.stack([np.linspace(-1.5, 1.5 + 0.1*q, K) for q in range(O)], axis=1)
    delta = centers[1] - centers[0]
    return ['const', FeatSpec('mult', idcs=[0, O - 1]), FeatSpec('mult', idcs=[1, 1, 0]),
            FeatSpec('rff', scale=np.sqrt(2./D), freq=freq, shift=2*np.pi*rng.random(D)), 'identity',
            FeatSpec('rbf', centers=centers, scale=-np.log(0.2)/delta**2, state_wise=state_wise)]


def structured_feats_np(feats):
    """ The same stack as objects of the NumPy restatement. """
    from oracle import pysim_oracle_np as onp
    from simurlacra_amd.specs import FeatSpec
    out = []
    for f in feats:
        if not isinstance(f, FeatSpec):
            out.append(f)
        elif f.kind == 'mult':
            out.append(onp.MultFeatNp(f.idcs))
        elif f.kind == 'rff':
            out.append(onp.RandFourierFeatNp(f.scale, f.freq, f.shift))
        else:
            out.append(onp.RBFFeatNp(f.centers, f.scale, f.state_wise))
    return out


def fnn_policy(kind, rng, hidden=(16, 16), nonlin='tanh', out_nonlin=None, scale=0.5, obs_dim=None):
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    ps = PolicySpec('fnn', O if obs_dim is None else obs_dim, A, hidden=list(hidden), hidden_nonlin=[nonlin]*len(hidden), out_nonlin=out_nonlin)
    return ps, scale*rng.standard_normal(ps.num_param)


def make_case(kind, n, seed, policy='linear', **pol_kwargs):
    rng = np.random.default_rng(seed)
    dp = domain_params(kind, n, rng)
    init = init_states(kind, dp, rng)
    ps, params = (linear_policy if policy == 'linear' else fnn_policy)(kind, rng, **pol_kwargs)
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    task = envdefs.default_task_spec(kind)
    return dict(kind=kind, dt=DT[kind], dp=dp, init=init, ps=ps, params=params, task=task, task_c=task.to_c(S, A))


def rel_err(a, b):
    """ max |a-b| / max(1, |b|) over finite reference entries; NaN placement must agree. """
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    fin = np.isfinite(b)
    assert np.array_equal(np.isnan(a), np.isnan(b)), 'NaN pattern differs'
    if not fin.any():
        return 0.
    return float(np.max(np.abs(a[fin] - b[fin])/np.maximum(1., np.abs(b[fin]))))


# wrapper chains (outermost first) exercised against both restatements and the kernels
WRAP_CASES = [
    # qbb_spota_hc-style stack: DomainRand(ActDelay(ActNorm(GaussianObsNoise(sim)))) with a randomized delay
    (abi.ENV_QBB, WrapSpec(ops=['act_delay', 'act_norm', 'obs_noise'], act_delay_row=20, noise_seed=5,
                           obs_noise_std=[0.01]*8, obs_noise_mean=[0.]*8)),
    (abi.ENV_QBB, WrapSpec(ops=['act_delay'], act_delay=2)),
    (abi.ENV_QQ, WrapSpec(ops=['obs_norm', 'act_norm'], obs_lo=[np.nan]*4 + [-30., -40.], obs_hi=[np.nan]*4 + [30., 40.])),
    (abi.ENV_QQ, WrapSpec(ops=['act_noise', 'obs_partial', 'obs_noise'], obs_keep=[1, 1, 0, 1, 1, 0], noise_seed=9,
                          act_noise_std=[0.2], act_noise_mean=[0.05], obs_noise_std=[0.02]*6, obs_noise_mean=[0.01]*6)),
    (abi.ENV_BOB, WrapSpec(ops=['act_norm', 'act_delay', 'obs_norm'], act_delay=4)),
    (abi.ENV_QCP_ST, WrapSpec(ops=['obs_partial', 'act_delay', 'act_noise'], obs_keep=[1, 1, 1, 0, 1], act_delay=1,
                              noise_seed=3, act_noise_std=[0.5])),
    (abi.ENV_OMO, WrapSpec(ops=['obs_noise', 'obs_norm', 'act_norm'], noise_seed=1, obs_noise_std=[0.1, 0.2])),
]




def custom_tasks():
    """ Tasks other than the environments' defaults, as a user passes them through `task_args`: dense cost
    matrices, every reward function, radial wrapping with other moduli, every FinalRewMode the reference implements
    (Pyrado/pyrado/tasks/final_reward.py:87-183).  Yields (kind, TaskSpec, keyword arguments of the NumPy Task). """
    from simurlacra_amd.specs import TaskSpec
    rng = np.random.default_rng(99)

    def dense(n, scale):
        m = rng.standard_normal((n, n))
        return scale*(m @ m.T + 0.3*rng.standard_normal((n, n)))   # not symmetric on purpose: e^T Q e is taken as written

    out = []
    Q, R = dense(4, 10.), dense(1, 0.1)
    out.append((abi.ENV_BOB, TaskSpec(np.array([0.1, 0., 0., 0.]), 'quadr', Q, R,
                                      final_mode=abi.FIN_ENABLED | abi.FIN_STATE_DEPENDENT, final_factor=50.),
                dict(state_des=[0.1, 0., 0., 0.], rew_kind='quadr', Q=Q, R=R, final_mode=dict(state_dependent=True),
                     final_factor=50.)))
    Q, R = np.diag([0.5, 2., 0.01, 0.02]), np.diag([0.01])
    out.append((abi.ENV_QQ, TaskSpec(np.array([0.2, np.pi, 0., 0.]), 'scaledexp', Q, R, radial_idcs=[0, 1], mod=[2*np.pi, np.pi],
                                     min_rew=1e-3, final_mode=abi.FIN_ENABLED | abi.FIN_TIME_DEPENDENT),
                dict(state_des=[0.2, np.pi, 0., 0.], rew_kind='scaledexp', Q=Q, R=R, radial_idcs=[0, 1],
                     mod=np.array([2*np.pi, np.pi]), min_rew=1e-3, final_mode=dict(time_dependent=True))))
    Q, R = dense(4, 0.5), np.diag([1e-2])
    out.append((abi.ENV_QCP_ST, TaskSpec(np.array([0., np.pi, 0., 0.]), 'exp', Q, R,
                                         final_mode=abi.FIN_ENABLED | abi.FIN_ALWAYS_POSITIVE),
                dict(state_des=[0., np.pi, 0., 0.], rew_kind='exp', Q=Q, R=R, final_mode=dict(always_positive=True))))
    Q, R = np.diag([3., 0.1]), np.diag([1e-3])
    out.append((abi.ENV_OMO, TaskSpec(np.array([0.3, 0.]), 'quadr', Q, R, radial_idcs=[0], mod=[0.5],
                                      final_mode=abi.FIN_ENABLED | abi.FIN_STATE_DEPENDENT | abi.FIN_TIME_DEPENDENT),
                dict(state_des=[0.3, 0.], rew_kind='quadr', Q=Q, R=R, radial_idcs=[0], mod=np.array([0.5]),
                     final_mode=dict(state_dependent=True, time_dependent=True))))
    Q, R = dense(8, 1.), dense(2, 0.05)
    out.append((abi.ENV_QBB, TaskSpec(0.01*np.arange(8.), 'quadr', Q, R,
                                      final_mode=abi.FIN_ENABLED | abi.FIN_ALWAYS_NEGATIVE | abi.FIN_STATE_DEPENDENT,
                                      final_factor=7.),
                dict(state_des=0.01*np.arange(8.), rew_kind='quadr', Q=Q, R=R,
                     final_mode=dict(always_negative=True, state_dependent=True), final_factor=7.)))
    sw = (2., 0.5, 1e-2, 0.5, 0.2)
    out.append((abi.ENV_QCP_SU, TaskSpec(np.array([0.1, np.pi, 0., 0.]), 'swingup', radial_idcs=[1], swingup=sw,
                                         final_mode=abi.FIN_ENABLED, final_factor=200.),
                dict(state_des=[0.1, np.pi, 0., 0.], rew_kind='swingup', radial_idcs=[1], swingup=sw, final_mode=dict(),
                     final_factor=200.)))
    return out


def custom_task_case(kind, n, seed, policy):
    """ Inputs for a custom-task test: a LinearPolicy (const, identity, squared features) or FNNPolicy whose gain
    makes a part of the rollouts leave the state space, so that the final rewards get exercised.  Returns the case
    and its horizon. """
    T = {abi.ENV_QQ: 150, abi.ENV_QCP_SU: 400, abi.ENV_QBB: 40}.get(kind, 60)
    if policy == 'fnn':
        case = make_case(kind, n, seed, policy='fnn', hidden=(32, 32), scale=1.0)
    else:
        # QBallBalancer's dead zone makes its closed loop discontinuous: with a high gain a one-ulp perturbation of
        # the initial state changes the oracle's own rewards by 1e-6 and more within 60 steps
        case = make_case(kind, n, seed, policy='linear', scale=0.3 if kind == abi.ENV_QBB else 8.0)
    if kind == abi.ENV_QCP_SU and policy == 'linear':   # a constant 8 V drives the cart off the rail
        case['params'][0] = 8.
    return case, T
