"""
CPU tests of the oracle (test infrastructure): the reference's own known-answer tests for this path, hand-derivable
anchors from the reference source, agreement of the two independent restatements (C and NumPy), and the committed
golden vectors.
"""
import glob
import os

import numpy as np
import pytest

from oracle import oracle_lib as orc, pysim_oracle_np as onp
from simurlacra_amd import _abi as abi, envdefs
from simurlacra_amd.specs import TaskSpec, PolicySpec, WrapSpec
from tests import cases

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


# ------------------------------------------------------------------ known-answer tests held by the reference
def test_kat_modulated_rew_fcn_numpy():
    """ Pyrado/tests/test_tasks.py:44-61 """
    s, a = np.array([1., 2., 3., 4.]), np.array([0., 0.])
    t = onp.Task(np.zeros(4), 'quadr', np.eye(4), np.eye(2), radial_idcs=[0, 1, 3], mod=2)
    assert t.step_rew(s, a) == -(1**2 + 3**2)
    t = onp.Task(np.zeros(4), 'quadr', np.eye(4), np.eye(2), radial_idcs=[1, 3], mod=np.array([2, 3]))
    assert t.step_rew(s, a) == -(1**2 + 3**2 + 1**2)


def test_kat_modulated_rew_fcn_c():
    """ Same numbers through the C restatement: the 4-dim case embedded in the 8-dim QBB state (zeros elsewhere). """
    s = np.array([1., 2., 3., 4., 0., 0., 0., 0.])
    a = np.zeros(2)
    dp = np.array(list(envdefs.nominal_domain_param(abi.ENV_QBB).values()))
    t = TaskSpec(np.zeros(8), 'quadr', np.eye(8), np.eye(2), radial_idcs=[0, 1, 3], mod=[2, 2, 2])
    assert orc.step_rew(abi.ENV_QBB, t.to_c(8, 2), dp, s, a) == -(1**2 + 3**2)
    t = TaskSpec(np.zeros(8), 'quadr', np.eye(8), np.eye(2), radial_idcs=[1, 3], mod=[2, 3])
    assert orc.step_rew(abi.ENV_QBB, t.to_c(8, 2), dp, s, a) == -(1**2 + 3**2 + 1**2)


def test_kat_final_reward_yields_once():
    """ FinalRewTask gives its reward once per reset (Pyrado/tests/test_tasks.py:181-188,252-259), here on the failure
    branch of OneMassOscillatorSim's task: -1000 when the state leaves the box, nothing afterwards. """
    kind = abi.ENV_OMO
    dp = np.array(list(envdefs.nominal_domain_param(kind).values()))[:, None]
    task_c = envdefs.default_task_spec(kind).to_c(2, 1)
    mem = orc.reset(kind, 0.01, 1000, task_c, np.array([[0.999], [9.0]]), dp)
    act = np.zeros((1, 1))
    _, r1, d1, st1, _ = orc.step(kind, 0.01, 1000, task_c, mem, act, dp)
    assert d1[0] == 1 and st1[0] & abi.ST_FAILED
    step_only = -(10*0.999**2 + 0.01*81.)
    assert r1[0] == pytest.approx(step_only - 1000., rel=1e-14)
    s_before = mem['state'].copy()
    _, r2, d2, _, _ = orc.step(kind, 0.01, 1000, task_c, mem, act, dp)
    assert d2[0] == 1
    assert r2[0] == pytest.approx(-(10*s_before[0, 0]**2 + 0.01*s_before[1, 0]**2), rel=1e-14)  # no second -1000


# ------------------------------------------------------------------------ anchors derivable from the source
def test_anchor_qq_constants():
    """ quanser_qube.py:62-90 with the nominal parameters """
    dp = np.array(list(envdefs.nominal_domain_param(abi.ENV_QQ).values()))
    c = orc.constants(abi.ENV_QQ, dp)[0]
    Mr, Lr, Mp, Lp, g = 0.095, 0.085, 0.024, 0.129, 9.81
    exp = [Mr*Lr**2/12 + Mp*Lr**2, 0.25*Mp*Lp**2, 0.5*Mp*Lp*Lr, Mp*Lp**2/12 + 0.25*Mp*Lp**2, 0.5*Mp*Lp*g]
    np.testing.assert_allclose(c[:5], exp, rtol=1e-15)
    np.testing.assert_allclose(c[:5], [2.30597916e-4, 9.9846e-5, 1.3158e-4, 1.33128e-4, 1.518588e-2], rtol=1e-6)


def test_anchor_omo_one_euler_step():
    """ one_mass_oscillator.py:75-84: from [-0.7, 0], u = 0, nominal (m=1, k=30, d=0.5), dt = 0.01 -> [-0.7, 0.21] """
    kind = abi.ENV_OMO
    dp = np.array([[1.], [30.], [0.5]])
    task_c = envdefs.default_task_spec(kind).to_c(2, 1)
    mem = orc.reset(kind, 0.01, 100, task_c, np.array([[-0.7], [0.]]), dp)
    obs, rew, done, st, _ = orc.step(kind, 0.01, 100, task_c, mem, np.zeros((1, 1)), dp)
    np.testing.assert_allclose(obs[:, 0], [-0.7, 0.21], rtol=1e-13)
    assert rew[0] == pytest.approx(-(10*0.49), rel=1e-14) and done[0] == 0


@pytest.mark.parametrize('kind', [abi.ENV_QQ, abi.ENV_OMO])
def test_anchor_zero_state_zero_action_is_a_fixed_point(kind):
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    dp = np.array(list(envdefs.nominal_domain_param(kind).values()))[:, None]
    task_c = envdefs.default_task_spec(kind).to_c(S, A)
    mem = orc.reset(kind, 0.01, 100, task_c, np.zeros((S, 1)), dp)
    orc.step(kind, 0.01, 100, task_c, mem, np.zeros((A, 1)), dp)
    assert np.all(mem['state'] == 0.)


def test_anchor_qq_rk4_uses_unadvanced_accelerations():
    """ quanser_qube.py:128-142: with the accelerations frozen at the start state, one step from s gives
    s + dt*[thd + dt/2*thdd, ald + dt/2*aldd, thdd, aldd] exactly (sum of the four stages). """
    kind, dt = abi.ENV_QQ, 0.004
    dp = np.array(list(envdefs.nominal_domain_param(kind).values()))
    task_c = envdefs.default_task_spec(kind).to_c(4, 1)
    s0 = np.array([0.3, -0.5, 1.1, -2.0])
    env = onp.NpEnv('qq', dt, 10)
    env.reset(s0.copy())
    c = env.c
    th, al, thd, ald = s0
    u = 1.5
    a = c[0] + c[1]*np.sin(al)**2; b = c[2]*np.cos(al); det = a*c[3] - b*b
    trq = dp[2]*(u - dp[2]*thd)/dp[1]
    c0 = c[1]*np.sin(2*al)*thd*ald - c[2]*np.sin(al)*ald*ald
    c1 = -0.5*c[1]*np.sin(2*al)*thd*thd + c[4]*np.sin(al)
    x = trq - dp[5]*thd - c0; y = -dp[8]*ald - c1
    thdd, aldd = (c[3]*x - b*y)/det, (a*y - b*x)/det
    expect = s0 + dt*np.array([thd + dt/2*thdd, ald + dt/2*aldd, thdd, aldd])
    mem = orc.reset(kind, dt, 10, task_c, s0[:, None], dp[:, None])
    orc.step(kind, dt, 10, task_c, mem, np.array([[u]]), dp[:, None])
    np.testing.assert_allclose(mem['state'][:, 0], expect, rtol=1e-13)


def test_qbb_dead_zone_is_written_into_in_bounds_actions_only():
    """ spaces/box.py:144-148 + quanser_ball_balancer.py:212-215: an action inside its bounds is the array the
    dynamics zero in place; a clipped action is a copy, so the caller's array keeps the raw value. """
    kind = abi.ENV_QBB
    dp = np.array(list(envdefs.nominal_domain_param(kind).values()))[:, None].repeat(2, axis=1)
    task_c = envdefs.default_task_spec(kind).to_c(8, 2)
    init = np.array([[0.1, 0.1], [0., 0.], [0., 0.], [0., 0.]])
    mem = orc.reset(kind, 0.01, 100, task_c, init, dp)
    act = np.array([[0.2, 0.2], [-0.05, 5.0]])  # env 0: both in the dead zone; env 1: second component out of bounds
    _, _, _, _, a_out = orc.step(kind, 0.01, 100, task_c, mem, act, dp)
    np.testing.assert_array_equal(a_out[:, 0], [0., 0.])
    np.testing.assert_array_equal(a_out[:, 1], [0.2, 5.0])


def test_qbb_ik_c_vs_numpy():
    for th in np.linspace(-np.pi/2.1, np.pi/2.1, 9):
        a, b = orc.qbb_ik(th, 0.0254, 0.275), onp.qbb_ik(th, 0.0254, 0.275)
        assert np.isfinite(a) and a == pytest.approx(b, abs=1e-13)


# ----------------------------------------------------------------------------------- C vs NumPy restatement
@pytest.mark.parametrize('kind', [abi.ENV_QQ, abi.ENV_QBB, abi.ENV_OMO, abi.ENV_QCP_ST, abi.ENV_BOB])
def test_infinite_state_components_give_nan_rewards_in_both_restatements(kind):
    """ The quadratic cost is err.dot(Q).dot(err) (rewards.py:67-72): an infinite error component meets the zeros of Q's
    other rows, and 0 * inf is a NaN -- in NumPy's products as in the C restatement's full sums.  The kernels for the
    default (diagonal) task are held to this in tests/test_gpu_parity.py::test_nan_and_infinite_initial_states.
    OneMassOscillator's dynamics are A.dot(state) + B.dot(act) (one_mass_oscillator.py:80-85): an infinite position makes
    the new position a NaN the same way. """
    name = cases.KIND_NAMES[kind]
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    nom = envdefs.nominal_domain_param(kind)
    dp = np.array(list(nom.values()), dtype=float)
    task_c = envdefs.default_task_spec(kind).to_c(S, A)
    for comp in range(S):
        for bad in (np.inf, -np.inf):
            state = 0.01*np.arange(1, S + 1, dtype=float)
            state[comp] = bad
            env = onp.NpEnv(name, cases.DT[kind], 10)
            env.reset(0.01*np.arange(1, (4 if kind == abi.ENV_QBB else S) + 1, dtype=float), dict(nom))
            with np.errstate(all='ignore'):
                r_np = env.task.step_rew(state.copy(), np.zeros(A))
            r_c = orc.step_rew(kind, task_c, dp, state, np.zeros(A))
            assert np.isnan(r_np) and np.isnan(r_c), (name, comp, bad, r_np, r_c)
    if kind == abi.ENV_OMO:
        env = onp.NpEnv(name, cases.DT[kind], 10)
        env.reset(np.array([0.1, 0.2]), dict(nom))
        env.state = np.array([np.inf, 0.2])
        with np.errstate(all='ignore'):
            env._dynamics(np.zeros(1))
        assert np.isnan(env.state[0]) and np.isinf(env.state[1])
        mem = orc.reset(kind, cases.DT[kind], 10, task_c, np.array([[np.inf], [0.2]]), dp.reshape(-1, 1))
        obs, rew, done, status, _ = orc.step(kind, cases.DT[kind], 10, task_c, mem, np.zeros((1, 1)), dp.reshape(-1, 1))
        assert np.isnan(mem['state'][0, 0]) and np.isinf(mem['state'][1, 0]) and done[0] and (status[0] & abi.ST_NAN) and np.isnan(rew[0])


@pytest.mark.parametrize('kind', cases.KINDS)
@pytest.mark.parametrize('pol', ['linear', 'fnn'])
def test_c_and_numpy_restatements_agree(kind, pol):
    case = cases.make_case(kind, 5, seed=31 + kind, policy=pol)
    T = 80
    ps, params = case['ps'], case['params']
    out = orc.rollout(kind, case['dt'], T, case['task_c'], ps, params, case['init'], case['dp'], record=True)
    keys = list(envdefs.nominal_domain_param(kind).keys())
    for i in range(5):
        env = onp.NpEnv(cases.KIND_NAMES[kind], case['dt'], T)
        policy = (onp.LinearPolicyNp(ps.feats, params.reshape(ps.act_dim, -1)) if pol == 'linear'
                  else onp.FNNPolicyNp(ps.obs_dim, ps.hidden, ps.act_dim, params))
        ro = onp.rollout_np(env, policy, case['init'][:, i], dict(zip(keys, case['dp'][:, i])))
        L = ro['length']
        assert L == out['length'][i]
        assert cases.rel_err(out['obs'][:L + 1, :, i], ro['observations']) < 1e-13
        assert cases.rel_err(out['act'][:L, :, i], ro['actions']) < 1e-13
        assert cases.rel_err(out['rew'][:L, i], ro['rewards']) < 1e-13
        assert out['ret'][i] == pytest.approx(ro['ret'], rel=1e-13, abs=1e-13)


def test_c_and_numpy_agree_with_action_delay_on_qbb():
    """ ActDelayWrapper queues array objects, so the QBB dead zone lands in the action recorded `delay` steps ago. """
    kind, T, delay = abi.ENV_QBB, 30, 3
    case = cases.make_case(kind, 3, seed=77, policy='linear', scale=0.2)
    dp = np.vstack([case['dp'], np.full((1, 3), float(delay))])
    out = orc.rollout(kind, case['dt'], T, case['task_c'], case['ps'], case['params'], case['init'], dp, record=True,
                      wrap=WrapSpec(ops=['act_delay'], act_delay_row=20))
    keys = list(envdefs.nominal_domain_param(kind).keys())
    for i in range(3):
        env = onp.NpEnv('qbb', case['dt'], T)
        policy = onp.LinearPolicyNp(case['ps'].feats, case['params'].reshape(2, -1))
        ro = onp.rollout_np(env, policy, case['init'][:, i], dict(zip(keys, case['dp'][:, i])), act_delay=delay)
        L = ro['length']
        assert L == out['length'][i]
        assert cases.rel_err(out['act'][:L, :, i], ro['actions']) < 1e-13
        assert cases.rel_err(out['rew'][:L, i], ro['rewards']) < 1e-13


NP_KIND = {abi.ENV_QQ: 'qq', abi.ENV_QCP_SU: 'qcp-su', abi.ENV_QCP_ST: 'qcp-st', abi.ENV_QBB: 'qbb', abi.ENV_BOB: 'bob',
           abi.ENV_OMO: 'omo', abi.ENV_PEND: 'pend', abi.ENV_QQ_ST: 'qq-st'}


def np_wrapped_env(kind, dt, T, wrap: WrapSpec, env_id):
    """ The NumPy restatement's env with the wrappers of `wrap` nested like the reference's objects. """
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    env = onp.NpEnv(NP_KIND[kind], dt, T)

    def vec(v, n):
        return np.zeros(n) if v is None else np.asarray(v, float)
    for name in reversed(list(wrap.ops)):  # innermost first
        if name == 'act_norm':
            env = onp.NpActNorm(env)
        elif name == 'act_delay':
            env = onp.NpActDelay(env, wrap.act_delay)
        elif name == 'act_noise':
            env = onp.NpActNoise(env, vec(wrap.act_noise_mean, A), vec(wrap.act_noise_std, A), wrap.noise_seed, env_id)
        elif name == 'obs_noise':
            keep = env.obs_bounds()[0].shape[0]
            # the product indexes noise vectors by the simulation's observation index
            assert keep == O, 'obs_noise outside obs_partial is expressed with full-length vectors'
            env = onp.NpObsNoise(env, vec(wrap.obs_noise_mean, O), vec(wrap.obs_noise_std, O), wrap.noise_seed, env_id)
        elif name == 'obs_norm':
            assert env.obs_bounds()[0].shape[0] == O
            env = onp.NpObsNorm(env, None if wrap.obs_lo is None else np.asarray(wrap.obs_lo, float),
                                None if wrap.obs_hi is None else np.asarray(wrap.obs_hi, float))
        elif name == 'obs_partial':
            env = onp.NpObsPartial(env, wrap.obs_keep)
    return env


WRAP_CASES = cases.WRAP_CASES


@pytest.mark.parametrize('kind,wrap', WRAP_CASES, ids=lambda v: '-'.join(v.ops) if isinstance(v, WrapSpec) else str(v))
def test_c_and_numpy_agree_on_wrapper_chains(kind, wrap):
    """ The fused wrapper chain of the C restatement against the NumPy one, where the wrappers are nested objects as
    in the reference (environment_wrappers/base.py:145-231): same recorded observations, actions, rewards. """
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    n, T = 4, 25
    O_pol = wrap.obs_dim(O)
    case = cases.make_case(kind, n, seed=5 + kind, policy='linear', scale=0.3, obs_dim=O_pol)
    dp = case['dp']
    delays = None
    if wrap.act_delay_row >= 0:
        delays = np.array([0, 1, 3, 2][:n], dtype=float)
        dp = np.vstack([dp, delays[None, :]])
    wrap.env_offset = 1000
    out = orc.rollout(kind, case['dt'], T, case['task_c'], case['ps'], case['params'], case['init'], dp, record=True, wrap=wrap)
    assert out['obs'].shape[1] == O_pol
    keys = list(envdefs.nominal_domain_param(kind).keys())
    for i in range(n):
        env = np_wrapped_env(kind, case['dt'], T, wrap, 1000 + i)
        policy = onp.LinearPolicyNp(case['ps'].feats, case['params'].reshape(A, -1))
        dpi = dict(zip(keys, case['dp'][:, i]))
        if delays is not None:
            dpi['act_delay'] = delays[i]
        ro = onp.rollout_np(env, policy, case['init'][:, i], dpi)
        L = ro['length']
        assert L == out['length'][i]
        assert cases.rel_err(out['obs'][:L + 1, :, i], ro['observations']) < 1e-12
        assert cases.rel_err(out['act'][:L, :, i], ro['actions']) < 1e-12
        assert cases.rel_err(out['rew'][:L, i], ro['rewards']) < 1e-12


@pytest.mark.parametrize('expl', ['normal', 'uniform'])
def test_c_and_numpy_agree_with_action_noise_exploration(expl):
    """ NormalActNoiseExplStrat / UniformActNoiseExplStrat (exploration/stochastic_action.py:42-68,96-176): the
    exploratory action is recorded and applied; here also delayed and normalised by the wrappers behind it. """
    import dataclasses
    kind, n, T = abi.ENV_QBB, 3, 25
    case = cases.make_case(kind, n, seed=21, policy='linear', scale=0.2)
    ps = dataclasses.replace(case['ps'], expl_kind=expl, expl_mean=[0.1, -0.2], expl_scale=[0.3, 0.5])
    wrap = WrapSpec(ops=['act_delay', 'act_norm'], act_delay=2, noise_seed=17, env_offset=40)
    out = orc.rollout(kind, case['dt'], T, case['task_c'], ps, case['params'], case['init'], case['dp'], record=True, wrap=wrap)
    keys = list(envdefs.nominal_domain_param(kind).keys())
    for i in range(n):
        env = np_wrapped_env(kind, case['dt'], T, wrap, 40 + i)
        policy = onp.ExplPolicyNp(onp.LinearPolicyNp(ps.feats, case['params'].reshape(2, -1)), expl, [0.1, -0.2], [0.3, 0.5],
                                  17, 40 + i)
        ro = onp.rollout_np(env, policy, case['init'][:, i], dict(zip(keys, case['dp'][:, i])))
        L = ro['length']
        assert L == out['length'][i]
        assert cases.rel_err(out['act'][:L, :, i], ro['actions']) < 1e-12
        assert cases.rel_err(out['rew'][:L, i], ro['rewards']) < 1e-12
    # the noise has the requested moments
    big = cases.make_case(abi.ENV_OMO, 4000, seed=2, policy='linear', feats=('const',), scale=0.0)
    ps = dataclasses.replace(big['ps'], expl_kind=expl, expl_mean=[0.5], expl_scale=[2.0])
    o = orc.rollout(abi.ENV_OMO, big['dt'], 1, big['task_c'], ps, big['params'], big['init'], big['dp'], record=True,
                    wrap=WrapSpec(noise_seed=3))
    a = o['act'][0, 0]
    assert abs(a.mean() - 0.5) < 0.1 and abs(a.std() - (2.0 if expl == 'normal' else 2.0/np.sqrt(3))) < 0.1
    assert expl == 'normal' or (a.min() >= -1.5 and a.max() < 2.5)


def test_c_and_numpy_agree_with_bernoulli_reset():
    """ rollout(bernoulli_reset=p), sampling/rollout.py:248-254: after each step the rollout stops with probability p.
    The lengths follow a geometric law truncated at max_steps. """
    kind, n, T, p = abi.ENV_OMO, 6, 40, 0.1
    case = cases.make_case(kind, n, seed=9, policy='linear', scale=0.2)
    wrap = WrapSpec(noise_seed=5, env_offset=70)
    out = orc.rollout(kind, case['dt'], T, case['task_c'], case['ps'], case['params'], case['init'], case['dp'], record=True,
                      wrap=wrap, bernoulli_reset=p)
    keys = list(envdefs.nominal_domain_param(kind).keys())
    for i in range(n):
        env = onp.NpEnv('omo', case['dt'], T)
        policy = onp.LinearPolicyNp(case['ps'].feats, case['params'].reshape(1, -1))
        ro = onp.rollout_np(env, policy, case['init'][:, i], dict(zip(keys, case['dp'][:, i])), bernoulli_reset=p,
                            seed=5, env_id=70 + i)
        L = ro['length']
        assert L == out['length'][i]
        assert bool(out['status'][i] & abi.ST_BERNOULLI) == (L < T)
        assert cases.rel_err(out['obs'][:L + 1, :, i], ro['observations']) < 1e-13
    assert (out['length'] < T).any()
    big = cases.make_case(kind, 20000, seed=1, policy='linear', scale=0.0)
    o = orc.rollout(kind, big['dt'], 1000, big['task_c'], big['ps'], big['params'], big['init'], big['dp'],
                    wrap=WrapSpec(noise_seed=11), bernoulli_reset=0.05, flags=0)
    assert abs(o['length'].mean() - 20.) < 0.6          # mean of the geometric law, 1/p
    assert abs((o['length'] == 1).mean() - 0.05) < 0.01


@pytest.mark.parametrize('state_wise', [True, False])
def test_c_and_numpy_agree_on_structured_features(state_wise):
    """ MultFeat, RandFourierFeat, RBFFeat (policies/features.py:143-303) inside a LinearPolicy: the C restatement
    against the NumPy one, which follows the reference's tensor expressions. """
    kind, n, T = abi.ENV_QQ, 4, 40
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    rng = np.random.default_rng(3)
    feats = cases.structured_feats(O, rng, state_wise)
    case = cases.make_case(kind, n, seed=12, policy='linear', feats=feats, scale=0.3)
    ps = case['ps']
    assert ps.num_feat == 1 + 1 + 1 + 24 + O + 5*O and ps.num_param == A*ps.num_feat
    out = orc.rollout(kind, case['dt'], T, case['task_c'], ps, case['params'], case['init'], case['dp'], record=True)
    keys = list(envdefs.nominal_domain_param(kind).keys())
    for i in range(n):
        env = onp.NpEnv('qq', case['dt'], T)
        policy = onp.LinearPolicyNp(cases.structured_feats_np(feats), case['params'].reshape(A, -1))
        ro = onp.rollout_np(env, policy, case['init'][:, i], dict(zip(keys, case['dp'][:, i])))
        L = ro['length']
        assert L == out['length'][i]
        assert cases.rel_err(out['act'][:L, :, i], ro['actions']) < 1e-12
        assert cases.rel_err(out['obs'][:L + 1, :, i], ro['observations']) < 1e-12
    # far from every center the activations underflow: the norm falls below 1e-8 and the division is by one
    rbf = onp.RBFFeatNp(np.linspace(-1, 1, 4).reshape(4, 1), 50., True)
    assert np.all(rbf(np.array([30.])) == 0.)


@pytest.mark.parametrize('kind,spec,np_kw', cases.custom_tasks(), ids=lambda v: cases.KIND_NAMES.get(v, '') if isinstance(v, int) else '')
def test_c_and_numpy_agree_on_custom_tasks(kind, spec, np_kw):
    """ rewards, final rewards and radial wrapping for tasks other than the defaults; a gain high enough that a part
    of the rollouts fails, so that the final reward is exercised """
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    n = 60
    case, T = cases.custom_task_case(kind, n, seed=40 + kind, policy='linear')
    out = orc.rollout(kind, case['dt'], T, spec.to_c(S, A), case['ps'], case['params'], case['init'], case['dp'], record=True)
    assert (out['status'] & abi.ST_FAILED).any() or kind in (abi.ENV_QCP_ST, abi.ENV_QCP_SU)   # final rewards are reached
    keys = list(envdefs.nominal_domain_param(kind).keys())
    for i in range(n):
        env = onp.NpEnv(cases.KIND_NAMES[kind], case['dt'], T, task=onp.Task(**np_kw))
        policy = onp.LinearPolicyNp(case['ps'].feats, case['params'].reshape(A, -1))
        ro = onp.rollout_np(env, policy, case['init'][:, i], dict(zip(keys, case['dp'][:, i])))
        L = ro['length']
        assert L == out['length'][i]
        assert cases.rel_err(out['rew'][:L, i], ro['rewards']) < 1e-11


def test_c_and_numpy_agree_on_simplified_qbb_dynamics_and_infinite_horizon_steps():
    """ QBallBalancerSim(simplified_dyn=True) (pysim/quanser_ball_balancer.py:251-252) through rollouts, and
    SimPyEnv.step with max_steps = inf (pysim/base.py:179,194): remaining_steps = 0, no time-out done. """
    kind, n, T = abi.ENV_QBB, 4, 50
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    case = cases.make_case(kind, n, seed=61, policy='linear', scale=0.3)
    flags = abi.F_STOP_ON_DONE | abi.F_SIMPLIFIED_DYN
    out = orc.rollout(kind, case['dt'], T, case['task_c'], case['ps'], case['params'], case['init'], case['dp'], record=True, flags=flags)
    full = orc.rollout(kind, case['dt'], T, case['task_c'], case['ps'], case['params'], case['init'], case['dp'], record=True)
    assert cases.rel_err(out['obs'][:10], full['obs'][:10]) > 1e-6      # it is a different model
    keys = list(envdefs.nominal_domain_param(kind).keys())
    for i in range(n):
        env = onp.NpEnv('qbb', case['dt'], T, simplified_dyn=True)
        policy = onp.LinearPolicyNp(case['ps'].feats, case['params'].reshape(A, -1))
        ro = onp.rollout_np(env, policy, case['init'][:, i], dict(zip(keys, case['dp'][:, i])))
        L = ro['length']
        assert L == out['length'][i]
        assert cases.rel_err(out['obs'][:L + 1, :, i], ro['observations']) < 1e-12
        assert cases.rel_err(out['rew'][:L, i], ro['rewards']) < 1e-12
    # stepping past max_steps with the infinite-horizon flag: OMO's time-dependent final reward sees remaining = 0
    kind = abi.ENV_OMO
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    case = cases.make_case(kind, 1, seed=2, policy='linear')
    mem = orc.reset(kind, case['dt'], 5, case['task_c'], case['init'], case['dp'])
    env = onp.NpEnv('omo', case['dt'], 5)
    env.reset(case['init'][:, 0], dict(zip(envdefs.nominal_domain_param(kind).keys(), case['dp'][:, 0])))
    for k in range(9):
        act = np.array([[0.1*k]])
        o, r, d, s, _ = orc.step(kind, case['dt'], 5, case['task_c'], mem, act, case['dp'], flags=abi.F_MAX_STEPS_INF)
        o2, r2, d2, _ = env.step(act[:, 0].copy(), max_steps_inf=True)
        assert not d[0] and not d2 and mem['curr_step'][0] == k + 1
        assert cases.rel_err(o[:, 0], o2) < 1e-13 and abs(r[0] - r2) < 1e-13


NONLIN_CASES = [(['relu', 'relu'], None), (['sigmoid', 'sigmoid'], None), ([None, 'tanh'], 'tanh'),
                (['relu', 'sigmoid', 'tanh'], 'sigmoid'), (['tanh'], 'relu')]


@pytest.mark.parametrize('hidden_nl,out_nl', NONLIN_CASES, ids=str)
def test_c_and_numpy_agree_on_fnn_nonlinearities(hidden_nl, out_nl):
    """ FNN.forward with every hidden / output nonlinearity the kernels know (policies/fnn.py:129-151) """
    kind, n, T = abi.ENV_QCP_SU, 3, 40
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    hidden = [24, 16, 8][:len(hidden_nl)]
    case = cases.make_case(kind, n, seed=5, policy='fnn', hidden=hidden)
    ps = PolicySpec('fnn', O, A, hidden=hidden, hidden_nonlin=hidden_nl, out_nonlin=out_nl)
    out = orc.rollout(kind, case['dt'], T, case['task_c'], ps, case['params'], case['init'], case['dp'], record=True)
    keys = list(envdefs.nominal_domain_param(kind).keys())
    for i in range(n):
        env = onp.NpEnv('qcp-su', case['dt'], T)
        policy = onp.FNNPolicyNp(O, hidden, A, case['params'], hidden_nonlin=hidden_nl, out_nonlin=out_nl)
        ro = onp.rollout_np(env, policy, case['init'][:, i], dict(zip(keys, case['dp'][:, i])))
        L = ro['length']
        assert L == out['length'][i]
        assert cases.rel_err(out['act'][:L, :, i], ro['actions']) < 1e-12
        assert cases.rel_err(out['rew'][:L, i], ro['rewards']) < 1e-12


def test_oracle_policy_forward_matches_torch():
    """ FNN.forward / LinearPolicy.forward against torch.nn.Linear in fp64 (policies/fnn.py:129-151) """
    import torch
    rng = np.random.default_rng(3)
    ps = PolicySpec('fnn', 6, 1, hidden=[64, 64], hidden_nonlin=['tanh', 'tanh'])
    params = rng.standard_normal(ps.num_param)
    obs = rng.standard_normal(6)
    lin = [torch.nn.Linear(6, 64), torch.nn.Linear(64, 64), torch.nn.Linear(64, 1)]
    net = torch.nn.Sequential(lin[0], torch.nn.Tanh(), lin[1], torch.nn.Tanh(), lin[2]).double()
    torch.nn.utils.vector_to_parameters(torch.as_tensor(params), net.parameters())
    with torch.no_grad():
        ref = net(torch.as_tensor(obs)).numpy()
    np.testing.assert_allclose(orc.policy_forward(ps, params, obs), ref, rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize('nl', ['tanh', 'relu', 'sigmoid', None])
def test_non_finite_values_pass_the_activations_as_in_torch(nl):
    """ A NaN weight or observation has to reach the action (rollout() ends on it, sampling/rollout.py:190-201), and an
    infinite observation saturates tanh and sigmoid: torch.relu keeps a NaN (a comparison-and-select form of relu turns it
    into 0 and the NaN is lost), tanh(+-inf) = +-1, sigmoid(+-inf) = 1 / 0. """
    import torch
    act = {'tanh': torch.nn.Tanh, 'relu': torch.nn.ReLU, 'sigmoid': torch.nn.Sigmoid, None: torch.nn.Identity}[nl]
    rng = np.random.default_rng(8)
    ps = PolicySpec('fnn', 3, 2, hidden=[5, 4], hidden_nonlin=[nl, nl])
    net = torch.nn.Sequential(torch.nn.Linear(3, 5), act(), torch.nn.Linear(5, 4), act(), torch.nn.Linear(4, 2)).double()
    for case in ('finite', 'nan weight', 'nan obs', 'inf obs', '-inf obs'):
        params, obs = rng.standard_normal(ps.num_param), rng.standard_normal(3)
        if case == 'nan weight':
            params[7] = np.nan    # first layer, second unit
        elif case != 'finite':
            obs[1] = dict(nan=np.nan, inf=np.inf)[case.split()[0].lstrip('-')]*(-1 if case[0] == '-' else 1)
        torch.nn.utils.vector_to_parameters(torch.as_tensor(params.copy()), net.parameters())
        with torch.no_grad():
            ref = net(torch.as_tensor(obs)).numpy()
        got = orc.policy_forward(ps, params, obs)
        np.testing.assert_array_equal(np.isnan(got), np.isnan(ref), err_msg=case)
        if case in ('nan weight', 'nan obs'):
            assert np.isnan(ref).all(), case
        np.testing.assert_allclose(got[~np.isnan(ref)], ref[~np.isnan(ref)], rtol=1e-12, atol=1e-13, err_msg=case)


# ------------------------------------------------------------------------------------------- golden vectors
@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(GOLDEN, 'rollout_*.npz'))), ids=os.path.basename)
def test_c_oracle_reproduces_golden_vectors(path):
    g = np.load(path)
    kind, T = int(g['kind']), int(g['T'])
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    task_c = envdefs.default_task_spec(kind).to_c(S, A)
    if len(g['hidden']) == 0:
        ps = PolicySpec('linear', O, A, feats=[str(f) for f in g['feats']])
    else:
        ps = PolicySpec('fnn', O, A, hidden=[int(h) for h in g['hidden']], hidden_nonlin=['tanh']*len(g['hidden']))
    out = orc.rollout(kind, float(g['dt']), T, task_c, ps, g['params'], g['init'], g['dp'], record=True)
    np.testing.assert_array_equal(out['length'], g['length'])
    for k in ('obs', 'act', 'rew', 'ret'):
        assert cases.rel_err(out[k], g[k]) < 1e-13, k


# ------------------------------------------------------------------------------- the product's device RNG
def test_indices_outside_their_tables_are_not_followed():
    """ The library's rule for dp_index / policy_index entries outside their tables, restated in the C oracle. """
    kind, n = abi.ENV_BOB, 40
    case = cases.make_case(kind, n, seed=6, policy='linear')
    table = case['dp'][:, :8].copy()
    dp_index = (np.arange(n) % 8).astype(np.int32)
    ref = orc.rollout(kind, case['dt'], 20, case['task_c'], case['ps'], case['params'], case['init'], table, dp_index=dp_index.copy())
    dp_index[[3, 39]] = [8, -1]
    out = orc.rollout(kind, case['dt'], 20, case['task_c'], case['ps'], case['params'], case['init'], table, dp_index=dp_index)
    assert list(out['status'][[3, 39]]) == [abi.ST_BAD_INDEX]*2 and list(out['length'][[3, 39]]) == [0, 0]
    ok = np.setdiff1d(np.arange(n), [3, 39])
    np.testing.assert_array_equal(out['ret'][ok], ref['ret'][ok])


def test_philox_known_answers():
    """ Philox4x32-10 vectors of the Random123 distribution (kat_vectors) """
    assert orc.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert orc.philox([0xffffffff]*4, [0xffffffff]*2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert orc.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_randomize_distributions_and_clipping():
    n = 200000
    nominal = [1., 2., 3., 4.]
    distrs = [dict(kind=abi.DISTR_NORMAL, row=0, a=5., b=2., clip_lo=1.5),
              dict(kind=abi.DISTR_UNIFORM, row=1, a=3., b=1.),
              dict(kind=abi.DISTR_BERNOULLI, row=2, a=-1., b=7., c=0.7),
              dict(kind=abi.DISTR_UNIFORM, row=3, a=5., b=5., clip_lo=0, roundint=1)]
    x = orc.randomize(n, distrs, nominal, seed=9)
    assert x[0].min() == 1.5 and abs(np.mean(x[0][x[0] > 1.5]) - 5.) < 0.2 and abs(np.median(x[0]) - 5.) < 0.03
    assert 2. <= x[1].min() and x[1].max() < 4. and abs(x[1].mean() - 3.) < 0.01
    assert set(np.unique(x[2])) == {-1., 7.} and abs((x[2] == 7.).mean() - 0.7) < 0.01
    assert np.all(x[3] == np.round(x[3])) and x[3].min() == 0 and x[3].max() == 10
    # counter-based: env i gets the same value whatever the batch it is drawn in
    y = orc.randomize(1000, distrs, nominal, seed=9, env_offset=5000)
    np.testing.assert_array_equal(y, x[:, 5000:6000])


def test_sample_init_lies_in_each_init_space():
    for kind in cases.KINDS:
        rng = np.random.default_rng(kind)
        dp = cases.domain_params(kind, 2000, rng)
        x = orc.sample_init(kind, 2000, dp, seed=3)
        if kind == abi.ENV_QBB:
            r = np.hypot(x[0], x[1])
            assert np.all(r >= 0.75*dp[3]/2*(1 - 1e-12)) and np.all(r <= 0.8*dp[3]/2*(1 + 1e-12))
            assert np.all(np.abs(x[2:]) <= 0.025)
        elif kind == abi.ENV_BOB:
            assert np.all(np.abs(x[0]) >= 0.7*dp[4]/2*(1 - 1e-12)) and np.all(np.abs(x[0]) <= 0.8*dp[4]/2*(1 + 1e-12))
            assert 0.4 < (x[0] > 0).mean() < 0.6
        elif kind == abi.ENV_OMO:
            assert np.all((x[0] >= -0.75) & (x[0] <= -0.65) & (np.abs(x[1]) <= 0.1))
        elif kind == abi.ENV_QQ:
            assert np.all(np.abs(x) <= (np.array([2, 1, .5, .5])/180*np.pi)[:, None])
