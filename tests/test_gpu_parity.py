"""
GPU parity tests: the HIP kernels, called through the C ABI of libpysimhip.so, against the CPU oracle on the same
seeded inputs.  Tolerance (floating point, stated by the task): per-step values within 1e-9 relative
(|a-b| <= 1e-9 * max(1, |b|)) over short horizons; rollout lengths and status words must be identical.
"""
import numpy as np
import pytest

from simurlacra_amd import _abi as abi, envdefs
from tests import cases

pytestmark = pytest.mark.gpu

TOL = 1e-9


@pytest.fixture(scope='module')
def eng():
    from simurlacra_amd.engine import get_engine
    return get_engine(0)


@pytest.fixture(scope='module')
def orc():
    from oracle import oracle_lib
    return oracle_lib


def _compare_rollouts(eng, orc, case, T, **kw):
    ref = orc.rollout(case['kind'], case['dt'], T, case['task_c'], case['ps'], case['params'], case['init'], case['dp'],
                      record=True, **kw)
    out = eng.rollout(case['kind'], case['dt'], T, case['task_c'], case['ps'], case['params'], case['init'], case['dp'],
                      record=True, **kw)
    got = {k: v.cpu().numpy() for k, v in out.items()}
    np.testing.assert_array_equal(got['length'], ref['length'])
    np.testing.assert_array_equal(got['status'], ref['status'])
    errs = {k: cases.rel_err(got[k], ref[k]) for k in ('obs', 'act', 'rew', 'ret', 'final_state')}
    assert max(errs.values()) <= TOL, errs
    return errs


@pytest.mark.parametrize('kind', cases.KINDS)
def test_rollout_linear_policy(eng, orc, kind):
    case = cases.make_case(kind, 300, seed=11 + kind, policy='linear')
    _compare_rollouts(eng, orc, case, T=60)


@pytest.mark.parametrize('kind', cases.KINDS)
@pytest.mark.parametrize('hidden', [(16, 16), (64, 64), (24,), (32, 20, 8)])
def test_rollout_fnn_policy(eng, orc, kind, hidden):
    case = cases.make_case(kind, 200, seed=101 + kind, policy='fnn', hidden=hidden)
    # QQubeStabSim sits at an unstable equilibrium: with a random 64x64 policy the oracle's own response to a one-ulp
    # change of the initial state reaches 1.6e-10 after 50 steps (1e-11 after 30), so its horizon is shorter
    _compare_rollouts(eng, orc, case, T=30 if kind == abi.ENV_QQ_ST else 50)


@pytest.mark.parametrize('policy', ['linear', 'fnn'])
def test_rollout_qbb_simplified_dynamics(eng, orc, policy):
    """ QBallBalancerSim(simplified_dyn=True) (pysim/quanser_ball_balancer.py:251-252): no IK prologue, other ball model """
    kw = dict(scale=0.3) if policy == 'linear' else dict(hidden=(32, 32), scale=0.3)
    case = cases.make_case(abi.ENV_QBB, 300, seed=15, policy=policy, **kw)
    _compare_rollouts(eng, orc, case, T=50, flags=abi.F_STOP_ON_DONE | abi.F_SIMPLIFIED_DYN)


def test_step_with_infinite_horizon_flag(eng, orc):
    """ SimPyEnv.step with max_steps = inf (pysim/base.py:179,194): stepping on past max_steps, never a time-out """
    import torch
    kind = abi.ENV_OMO
    case = cases.make_case(kind, 70, seed=2, policy='linear')
    mem_ref = orc.reset(kind, case['dt'], 5, case['task_c'], case['init'], case['dp'])
    mem = eng.reset(kind, case['dt'], 5, case['task_c'], case['init'], case['dp'])
    for k in range(9):
        act = np.full((1, 70), 0.1*k)
        o_ref, r_ref, d_ref, s_ref, _ = orc.step(kind, case['dt'], 5, case['task_c'], mem_ref, act, case['dp'], flags=abi.F_MAX_STEPS_INF)
        o, r, d, s = eng.step(kind, case['dt'], 5, case['task_c'], mem, torch.as_tensor(act, device=eng.device), case['dp'],
                              flags=abi.F_MAX_STEPS_INF)
        np.testing.assert_array_equal(d.cpu().numpy(), d_ref)
        np.testing.assert_array_equal(s.cpu().numpy(), s_ref)
        assert not d_ref.any() and cases.rel_err(o.cpu().numpy(), o_ref) <= TOL and cases.rel_err(r.cpu().numpy(), r_ref) <= TOL


@pytest.mark.parametrize('policy', ['linear', 'fnn'])
@pytest.mark.parametrize('kind,spec,np_kw', cases.custom_tasks(), ids=lambda v: cases.KIND_NAMES.get(v, '') if isinstance(v, int) else '')
def test_rollout_with_custom_tasks(eng, orc, kind, spec, np_kw, policy):
    """ dense cost matrices, every reward function and FinalRewMode, radial wrapping with other moduli """
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    case, T = cases.custom_task_case(kind, 300, seed=70 + kind, policy=policy)
    case['task_c'] = spec.to_c(S, A)
    _compare_rollouts(eng, orc, case, T=T)


@pytest.mark.parametrize('hidden_nl,out_nl', [(['relu', 'relu'], None), (['sigmoid', 'sigmoid'], None), ([None, 'tanh'], 'tanh'),
                                              (['relu', 'sigmoid', 'tanh'], 'sigmoid'), (['tanh'], 'relu')], ids=str)
def test_rollout_fnn_nonlinearities(eng, orc, hidden_nl, out_nl):
    """ every hidden / output nonlinearity of FNN.forward (policies/fnn.py:129-151), mixed per layer """
    from simurlacra_amd.specs import PolicySpec
    for kind in (abi.ENV_QCP_SU, abi.ENV_QBB):
        S, O, A, P, _ = abi.ENV_DIMS[kind]
        hidden = [64, 32, 16][:len(hidden_nl)]
        case = cases.make_case(kind, 200, seed=8 + kind, policy='fnn', hidden=hidden)
        case['ps'] = PolicySpec('fnn', O, A, hidden=hidden, hidden_nonlin=hidden_nl, out_nonlin=out_nl)
        _compare_rollouts(eng, orc, case, T=40)


@pytest.mark.parametrize('kind', [abi.ENV_QQ, abi.ENV_QBB])
def test_rollout_deepest_widest_fnn(eng, orc, kind):
    """ 4 hidden layers of 64: the weight image leaves no LDS for the fine tanh table, the kernel variant with the
    coarse table runs. """
    # small weights and a short horizon: with a high-gain random policy the closed loop is chaotic and a one-ulp
    # difference (the oracle's own, under a perturbed initial state) grows past the tolerance within 20 steps
    case = cases.make_case(kind, 150, seed=31 + kind, policy='fnn', hidden=(64, 64, 64, 64), scale=0.15)
    _compare_rollouts(eng, orc, case, T=25)


@pytest.mark.parametrize('hidden', [(64, 64), (64, 64, 64, 64)])
def test_fnn_nan_parameter_is_flagged(eng, orc, hidden):
    """ A NaN weight must reach the action through the activation function (both tanh table variants). """
    kind = abi.ENV_QQ
    case = cases.make_case(kind, 70, seed=77, policy='fnn', hidden=hidden)
    params = case['params'].copy()
    params[5] = np.nan
    ref = orc.rollout(kind, case['dt'], 20, case['task_c'], case['ps'], params, case['init'], case['dp'])
    out = eng.rollout(kind, case['dt'], 20, case['task_c'], case['ps'], params, case['init'], case['dp'])
    np.testing.assert_array_equal(out['status'].cpu().numpy(), ref['status'])
    np.testing.assert_array_equal(out['length'].cpu().numpy(), ref['length'])
    assert np.all(out['status'].cpu().numpy() & abi.ST_NAN)


@pytest.mark.parametrize('policy', ['linear', 'fnn', 'fnn64'])
@pytest.mark.parametrize('kind,wrap', cases.WRAP_CASES, ids=lambda v: '-'.join(v.ops) if hasattr(v, 'ops') else str(v))
def test_rollout_with_fused_wrappers(eng, orc, kind, wrap, policy):
    """ ActNorm / ActDelay (fixed and per-env randomized) / Gaussian action and observation noise / ObsNorm /
    ObsPartial fused into the step loop, in the nesting order of the Python objects. """
    import copy
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    n, T = 300, 40
    wrap = copy.deepcopy(wrap)
    wrap.env_offset = 12345678901
    kw = dict(obs_dim=wrap.obs_dim(O), scale=0.3)
    if policy != 'linear':
        kw['hidden'] = (32, 32) if policy == 'fnn' else (64, 64)   # (64, 64): the kernel with the weights in registers
        if policy == 'fnn64':
            kw['scale'] = 0.1   # (QBallBalancer under a 64 x 64 net of larger weights amplifies rounding past the tolerance within 40 steps)
    case = cases.make_case(kind, n, seed=55 + kind, policy=policy[:3] if policy != 'linear' else policy, **kw)
    dp = case['dp']
    if wrap.act_delay_row >= 0:
        dp = np.vstack([dp, np.random.default_rng(3).integers(0, 6, n).astype(float)[None, :]])
    case = dict(case, dp=dp)
    _compare_rollouts(eng, orc, case, T=T, wrap=wrap)


@pytest.mark.parametrize('policy', ['linear', 'fnn', 'fnn64', 'population', 'population64'])
@pytest.mark.parametrize('kind,spec,np_kw', [c for c in cases.custom_tasks() if c[0] in (abi.ENV_QQ, abi.ENV_QBB, abi.ENV_BOB)],
                         ids=lambda v: cases.KIND_NAMES.get(v, '') if isinstance(v, int) else '')
def test_fused_wrappers_around_a_task_without_the_default_shape(eng, orc, kind, spec, np_kw, policy):
    """ The wrapped kernels exist twice: for the simulation's default task shape (decided at compile time) and for any
    other task.  The wrapper tests above run the former; here a dense-weight custom task forces the latter, for a
    LinearPolicy on the observations, FNNs from LDS and from registers, and populations of them. """
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    wrap = {abi.ENV_QQ: cases.WrapSpec(ops=['act_norm', 'act_delay'], act_delay=3),
            abi.ENV_QBB: cases.WrapSpec(ops=['act_delay', 'act_norm', 'obs_noise'], act_delay=2, noise_seed=7, env_offset=3,
                                        obs_noise_std=[0.01]*8),
            abi.ENV_BOB: cases.WrapSpec(ops=['act_noise', 'act_norm'], noise_seed=2, act_noise_std=[0.05])}[kind]
    R = 8
    n, T = (40*R, 30) if policy.startswith('population') else (300, 30)
    if policy == 'linear':
        case = cases.make_case(kind, n, seed=31 + kind, policy='linear', feats=['identity'], scale=0.3)
    else:
        case = cases.make_case(kind, n, seed=31 + kind, policy='fnn', hidden=(64, 64) if policy.endswith('64') else (32, 32),
                               scale=0.1)
    case['task_c'] = spec.to_c(S, A)
    kw = dict(wrap=wrap)
    if policy.startswith('population'):
        rng = np.random.default_rng(5)
        case['params'] = 0.1*rng.standard_normal((n//R, case['ps'].num_param))
        kw['policy_group'] = R
    _compare_rollouts(eng, orc, case, T=T, **kw)


@pytest.mark.parametrize('policy', ['linear', 'linear_simple', 'fnn', 'fnn64', 'population', 'population64'])
@pytest.mark.parametrize('kind', cases.KINDS, ids=[cases.KIND_NAMES[k] for k in cases.KINDS])
def test_a_lone_act_norm_wrapper(eng, orc, kind, policy):
    """ ActNormWrapper alone in front of the simulation (the usual configuration of the reference's training scripts) runs
    on the kernels of the bare simulation, which denormalise the action themselves: trajectories against the oracle, incl.
    QBallBalancer, whose dead zone then no longer reaches the recorded action (the wrapper hands the simulation a new array). """
    R = 8
    n, T = (30*R, 40) if policy.startswith('population') else (300, 40)
    if policy == 'linear':
        case = cases.make_case(kind, n, seed=91 + kind, policy='linear', scale=0.3)
    elif policy == 'linear_simple':
        case = cases.make_case(kind, n, seed=91 + kind, policy='linear', feats=['const', 'identity'], scale=0.3)
    else:
        case = cases.make_case(kind, n, seed=91 + kind, policy='fnn', hidden=(64, 64) if policy.endswith('64') else (32, 32),
                               scale=0.1)
    kw = dict(wrap=cases.WrapSpec(ops=['act_norm']))
    if policy.startswith('population'):
        case['params'] = 0.1*np.random.default_rng(6).standard_normal((n//R, case['ps'].num_param))
        kw['policy_group'] = R
    _compare_rollouts(eng, orc, case, T=T, **kw)


@pytest.mark.parametrize('kind', [abi.ENV_QQ, abi.ENV_QBB, abi.ENV_QCP_SU], ids=['qq', 'qbb', 'qcp-su'])
@pytest.mark.parametrize('n,hidden', [(300, (16, 16)), (9000, (32, 32)), (9000, (64, 64)), (9000, (64, 48, 32))], ids=str)
def test_a_lone_act_norm_wrapper_on_the_single_policy_fnn_kernels(eng, orc, kind, n, hidden):
    """ Launches that take the bare single-policy FNN kernels with the action denormalised inside (more environments than
    the population-of-copies layout of small launches serves, or a network it does not serve), recorded and compared in full. """
    case = cases.make_case(kind, n, seed=17 + kind, policy='fnn', hidden=hidden, scale=0.1)
    _compare_rollouts(eng, orc, case, T=30, wrap=cases.WrapSpec(ops=['act_norm']))


@pytest.mark.parametrize('expl', ['normal', 'uniform'])
@pytest.mark.parametrize('policy', ['linear', 'fnn'])
def test_rollout_with_action_noise_exploration(eng, orc, expl, policy):
    """ NormalActNoiseExplStrat / UniformActNoiseExplStrat fused behind the policy forward, alone and in front of a
    wrapper chain: same Philox draws as the oracle, so trajectories agree to the tolerance. """
    import dataclasses
    for kind, wrap in ((abi.ENV_QQ, cases.WrapSpec(noise_seed=8, env_offset=5)),
                       (abi.ENV_QBB, cases.WrapSpec(ops=['act_delay', 'act_norm', 'obs_noise'], act_delay=2, noise_seed=11,
                                                    obs_noise_std=[0.01]*8))):
        S, O, A, P, _ = abi.ENV_DIMS[kind]
        kw = dict(scale=0.3)
        if policy == 'fnn':
            kw['hidden'] = (64, 64)
        case = cases.make_case(kind, 300, seed=91 + kind, policy=policy, **kw)
        case['ps'] = dataclasses.replace(case['ps'], expl_kind=expl, expl_mean=[0.05]*A, expl_scale=[0.4]*A)
        _compare_rollouts(eng, orc, case, T=40, wrap=wrap)


@pytest.mark.parametrize('policy', ['linear', 'fnn'])
def test_rollout_with_bernoulli_reset(eng, orc, policy):
    """ rollout(bernoulli_reset=p) (sampling/rollout.py:248-254): stop with probability p after each step, alone and
    together with a wrapper chain; lanes freed early are refilled. """
    for kind, wrap, n in ((abi.ENV_QQ, cases.WrapSpec(noise_seed=21, env_offset=9), 300),
                          (abi.ENV_BOB, cases.WrapSpec(ops=['act_norm', 'obs_noise'], noise_seed=2, obs_noise_std=[0.01]*4), 200000)):
        kw = dict(scale=0.3)
        if policy == 'fnn':
            kw['hidden'] = (16, 16)
        case = cases.make_case(kind, n, seed=14 + kind, policy=policy, **kw)
        if n > 1000:
            ref = orc.rollout(kind, case['dt'], 40, case['task_c'], case['ps'], case['params'], case['init'], case['dp'],
                              wrap=wrap, bernoulli_reset=0.07)
            out = eng.rollout(kind, case['dt'], 40, case['task_c'], case['ps'], case['params'], case['init'], case['dp'],
                              wrap=wrap, bernoulli_reset=0.07)
            np.testing.assert_array_equal(out['length'].cpu().numpy(), ref['length'])
            np.testing.assert_array_equal(out['status'].cpu().numpy(), ref['status'])
            assert cases.rel_err(out['ret'].cpu().numpy(), ref['ret']) <= TOL
        else:
            _compare_rollouts(eng, orc, case, T=40, wrap=wrap, bernoulli_reset=0.07)
    with pytest.raises(Exception):
        eng.rollout(kind, case['dt'], 40, case['task_c'], case['ps'], case['params'], case['init'], case['dp'], bernoulli_reset=1.5)


@pytest.mark.parametrize('mode', ['shared', 'population', 'wrapped'])
def test_rollout_with_structured_features(eng, orc, mode):
    """ MultFeat, RandFourierFeat, RBFFeat (policies/features.py:143-303) evaluated inside the LinearPolicy kernel: one
    weight set in LDS, a population of weight sets in global memory, and behind an ObsPartialWrapper. """
    kind = abi.ENV_QQ
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    wrap = cases.WrapSpec(ops=['obs_partial', 'act_norm'], obs_keep=[1, 1, 1, 1, 0, 1]) if mode == 'wrapped' else None
    Oe = O if wrap is None else wrap.obs_dim(O)
    for state_wise in (True, False):
        feats = cases.structured_feats(Oe, np.random.default_rng(8), state_wise)
        n, R = (320, 16) if mode == 'population' else (300, 0)
        case = cases.make_case(kind, n, seed=3, policy='linear', feats=feats, scale=0.3, obs_dim=Oe)
        if mode == 'population':
            case['params'] = 0.3*np.random.default_rng(4).standard_normal((n//R, case['ps'].num_param))
        kw = dict(policy_group=R) if R else {}
        if wrap is not None:
            kw['wrap'] = wrap
        _compare_rollouts(eng, orc, case, T=40, **kw)
    # descriptors the library must reject
    import dataclasses
    from simurlacra_amd.specs import FeatSpec
    bad = dataclasses.replace(case['ps'], feats=[FeatSpec('mult', idcs=[0, 7])])   # QQube observes 6 values
    with pytest.raises(Exception):
        eng.rollout(kind, case['dt'], 5, case['task_c'], bad, np.zeros(1), case['init'], case['dp'])


def test_fused_delay_with_lane_refill(eng, orc):
    """ More envs than resident lanes and short episodes: lanes take over new envs, whose delay queues must start
    empty although the ring-buffer memory is reused. """
    kind = abi.ENV_BOB
    n = 300000
    case = cases.make_case(kind, n, seed=4, policy='linear', scale=2.0)
    wrap = cases.WrapSpec(ops=['act_delay', 'act_norm'], act_delay_row=8)
    dp = np.vstack([case['dp'], np.random.default_rng(1).integers(0, 9, n).astype(float)[None, :]])
    ref = orc.rollout(kind, case['dt'], 30, case['task_c'], case['ps'], case['params'], case['init'], dp, wrap=wrap)
    out = eng.rollout(kind, case['dt'], 30, case['task_c'], case['ps'], case['params'], case['init'], dp, wrap=wrap)
    np.testing.assert_array_equal(out['length'].cpu().numpy(), ref['length'])
    assert cases.rel_err(out['ret'].cpu().numpy(), ref['ret']) <= TOL


@pytest.mark.parametrize('policy', ['linear', 'fnn'])
@pytest.mark.parametrize('setup', ['bare', 'delay-alias', 'partial-obs', 'refill'])
def test_env_major_trajectories_hold_the_same_numbers(eng, setup, policy):
    """ PSH_F_TRAJ_ENV_MAJOR: obs [N, T+1, O'], act [N, T, A], rew [N, T] hold, bit for bit, what the time-major arrays
    hold (same kernels, other addresses), entries past an episode's end stay untouched in both; incl. the ActDelayWrapper
    case in which the simulation rewrites an action recorded `delay` steps earlier, a narrower observation, and a launch
    with more envs than resident lanes. """
    kind, n, T, wrap = abi.ENV_QBB, 300, 40, None
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    if setup == 'delay-alias':
        wrap = cases.WrapSpec(ops=['act_delay'], act_delay=3)
    elif setup == 'partial-obs':
        wrap = cases.WrapSpec(ops=['obs_partial', 'act_norm'], obs_keep=[1, 0, 1, 1, 1, 0, 1, 1])
    elif setup == 'refill':
        kind, n, T = abi.ENV_BOB, 300000, 100
        S, O, A, P, _ = abi.ENV_DIMS[kind]
    kw = dict(scale=0.5 if kind == abi.ENV_QBB else 2.0, obs_dim=O if wrap is None else wrap.obs_dim(O))
    if policy == 'fnn':
        kw['hidden'] = (16, 16)
    case = cases.make_case(kind, n, seed=77, policy=policy, **kw)
    args = (kind, case['dt'], T, case['task_c'], case['ps'], case['params'], case['init'], case['dp'])
    tm = eng.rollout(*args, record=True, wrap=wrap)
    em = eng.rollout(*args, record=True, wrap=wrap, flags=abi.F_STOP_ON_DONE | abi.F_TRAJ_ENV_MAJOR)
    Oe = kw['obs_dim']
    assert tuple(em['obs'].shape) == (n, T + 1, Oe) and tuple(em['act'].shape) == (n, T, A) and tuple(em['rew'].shape) == (n, T)
    for k in ('ret', 'length', 'status', 'final_state'):
        assert bool((tm[k] == em[k]).all()), k
    import torch
    for k, perm in (('obs', (1, 2, 0)), ('act', (1, 2, 0)), ('rew', (1, 0))):
        a, b = tm[k], em[k].permute(*perm)
        assert bool(((a == b) | (torch.isnan(a) & torch.isnan(b))).all()), k
    assert int(tm['length'].min()) < T or setup != 'refill'     # the refill case has episodes that end early


@pytest.mark.parametrize('kind', cases.KINDS)
def test_reset_step(eng, orc, kind):
    import torch
    case = cases.make_case(kind, 130, seed=7 + kind)
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    T = 40
    rng = np.random.default_rng(5)
    mem_ref = orc.reset(kind, case['dt'], T, case['task_c'], case['init'], case['dp'])
    mem = eng.reset(kind, case['dt'], T, case['task_c'], case['init'], case['dp'])
    assert cases.rel_err(mem['state'].cpu().numpy(), mem_ref['state']) <= TOL
    assert cases.rel_err(mem['obs'].cpu().numpy(), mem_ref['obs']) <= TOL
    _, alo, ahi = orc.constants(kind, case['dp'][:, 0])[2:5]
    for _ in range(T):
        act = rng.uniform(-1.5, 1.5, (A, 130))*ahi[:, None]
        o_ref, r_ref, d_ref, s_ref, a_ref = orc.step(kind, case['dt'], T, case['task_c'], mem_ref, act, case['dp'])
        act_t = torch.as_tensor(act, device=eng.device).contiguous()
        o, r, d, s = eng.step(kind, case['dt'], T, case['task_c'], mem, act_t, case['dp'])
        np.testing.assert_array_equal(d.cpu().numpy(), d_ref)
        np.testing.assert_array_equal(s.cpu().numpy(), s_ref)
        assert cases.rel_err(o.cpu().numpy(), o_ref) <= TOL
        assert cases.rel_err(r.cpu().numpy(), r_ref) <= TOL
        assert cases.rel_err(act_t.cpu().numpy(), a_ref) <= TOL
        np.testing.assert_array_equal(mem['curr_step'].cpu().numpy(), mem_ref['curr_step'])


import glob
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(GOLDEN, 'rollout_*.npz'))), ids=os.path.basename)
def test_golden_vectors(eng, path):
    """ The committed golden vectors (NumPy restatement of the reference, tests/golden/make_golden.py). """
    from simurlacra_amd import envdefs
    from simurlacra_amd.specs import PolicySpec
    g = np.load(path)
    kind, T = int(g['kind']), int(g['T'])
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    task_c = envdefs.default_task_spec(kind).to_c(S, A)
    if len(g['hidden']) == 0:
        ps = PolicySpec('linear', O, A, feats=[str(f) for f in g['feats']])
    else:
        ps = PolicySpec('fnn', O, A, hidden=[int(h) for h in g['hidden']], hidden_nonlin=['tanh']*len(g['hidden']))
    out = eng.rollout(kind, float(g['dt']), T, task_c, ps, g['params'], g['init'], g['dp'], record=True)
    np.testing.assert_array_equal(out['length'].cpu().numpy(), g['length'])
    for k in ('obs', 'act', 'rew', 'ret'):
        assert cases.rel_err(out[k].cpu().numpy(), g[k]) <= TOL, k


def test_randomize_matches_oracle(eng, orc):
    """ Same Philox counters on both sides: integer stream bit-exact, so Uniform/Bernoulli/rounded values are
    identical and Normal values differ only by the last bits of log/sqrt/cos. """
    nominal = [1., 2., 3., 4., 5.]
    distrs = [dict(kind=abi.DISTR_NORMAL, row=0, a=5., b=2., clip_lo=1.5),
              dict(kind=abi.DISTR_UNIFORM, row=1, a=3., b=1.),
              dict(kind=abi.DISTR_BERNOULLI, row=2, a=-1., b=7., c=0.7),
              dict(kind=abi.DISTR_UNIFORM, row=3, a=5., b=5., clip_lo=0, roundint=1)]
    n = 100000
    ref = orc.randomize(n, distrs, nominal, seed=42, env_offset=123456789012)
    got = eng.randomize(n, distrs, nominal, seed=42, env_offset=123456789012).cpu().numpy()
    np.testing.assert_array_equal(got[1:], ref[1:])
    assert cases.rel_err(got[0], ref[0]) <= 1e-13


@pytest.mark.parametrize('kind', cases.KINDS)
def test_sample_init_matches_oracle(eng, orc, kind):
    rng = np.random.default_rng(kind)
    dp = cases.domain_params(kind, 5000, rng)
    ref = orc.sample_init(kind, 5000, dp, seed=7, env_offset=99)
    got = eng.sample_init(kind, 5000, dp, seed=7, env_offset=99).cpu().numpy()
    assert cases.rel_err(got, ref) <= 1e-14


def test_qbb_ik_matches_oracle(eng, orc):
    th = np.linspace(-np.pi/2.1, np.pi/2.1, 50)
    r_arm = np.full(50, 0.0254)*np.linspace(0.8, 1.2, 50)
    l_plate = np.full(50, 0.275)*np.linspace(1.2, 0.8, 50)
    got = eng.qbb_ik(th, r_arm, l_plate).cpu().numpy()
    ref = np.array([orc.qbb_ik(a, b, c) for a, b, c in zip(th, r_arm, l_plate)])
    assert np.all(np.isfinite(got)) and cases.rel_err(got, ref) <= 1e-12


def test_segment_mean(eng):
    import torch
    ret = torch.randn(37*10, dtype=torch.float64, device=eng.device)
    got = eng.segment_mean(ret, 10).cpu().numpy()
    np.testing.assert_allclose(got, ret.view(37, 10).mean(dim=1).cpu().numpy(), rtol=1e-12, atol=1e-15)


def test_buffer_mode_and_population_mode(eng, orc):
    """ dp_index (DomainRandWrapperBuffer / sweep mode) and policy_index (one LinearPolicy row per env) """
    kind = abi.ENV_BOB
    case = cases.make_case(kind, 256, seed=5, policy='linear')
    rng = np.random.default_rng(1)
    table = case['dp'][:, :16].copy()
    dp_index = rng.integers(0, 16, 256).astype(np.int32)
    pop = 0.5*rng.standard_normal((8, case['ps'].num_param))
    policy_index = (np.arange(256)//32).astype(np.int32)
    kw = dict(dp_index=dp_index, policy_index=policy_index, record=True)
    ref = orc.rollout(kind, case['dt'], 50, case['task_c'], case['ps'], pop, case['init'], table, **kw)
    out = eng.rollout(kind, case['dt'], 50, case['task_c'], case['ps'], pop, case['init'], table, **kw)
    np.testing.assert_array_equal(out['length'].cpu().numpy(), ref['length'])
    for k in ('obs', 'act', 'rew', 'ret'):
        assert cases.rel_err(out[k].cpu().numpy(), ref[k]) <= TOL, k


@pytest.mark.parametrize('policy', ['linear', 'fnn'])
def test_indices_outside_their_tables_are_not_followed(eng, orc, policy):
    """ dp_index / policy_index entries that point outside the domain-parameter table or the policy rows: those envs
    are not run and say so in their status word, all others are untouched (no counterpart in the reference, whose
    Python lists raise IndexError; a kernel must not read there). """
    kind, n = abi.ENV_QQ, 300
    kw = dict(hidden=(16, 16)) if policy == 'fnn' else {}
    case = cases.make_case(kind, n, seed=6, policy=policy, scale=0.3, **kw)
    table = case['dp'][:, :16].copy()
    rng = np.random.default_rng(3)
    dp_index = rng.integers(0, 16, n).astype(np.int32)
    good = dict(dp_index=dp_index.copy())
    bad_envs = [0, 17, 63, 64, 299]
    dp_index[bad_envs] = [16, -1, 2**30, -2**31, 1000]
    kw = dict(dp_index=dp_index)
    params = case['params']
    if policy == 'linear':
        params = 0.3*rng.standard_normal((4, case['ps'].num_param))
        pol_index = rng.integers(0, 4, n).astype(np.int32)
        good['policy_index'] = pol_index.copy()
        pol_index[[5, 128]] = [4, -7]
        bad_envs += [5, 128]
        kw['policy_index'] = pol_index
    ref_good = orc.rollout(kind, case['dt'], 40, case['task_c'], case['ps'], params, case['init'], table, **good)
    ref = orc.rollout(kind, case['dt'], 40, case['task_c'], case['ps'], params, case['init'], table, **kw)
    out = eng.rollout(kind, case['dt'], 40, case['task_c'], case['ps'], params, case['init'], table, **kw)
    st, ln, ret = (out[k].cpu().numpy() for k in ('status', 'length', 'ret'))
    ok = np.setdiff1d(np.arange(n), bad_envs)
    assert np.all(st[bad_envs] == abi.ST_BAD_INDEX) and np.all(ln[bad_envs] == 0) and np.all(ret[bad_envs] == 0.)
    np.testing.assert_array_equal(st, ref['status'])
    np.testing.assert_array_equal(ln, ref['length'])
    np.testing.assert_array_equal(ln[ok], ref_good['length'][ok])
    assert cases.rel_err(ret, ref['ret']) <= TOL
    # reset / step and init sampling
    mem = eng.reset(kind, case['dt'], 40, case['task_c'], case['init'], table, dp_index=dp_index)
    st0 = mem['status'].cpu().numpy()
    assert np.all(st0[bad_envs[:5]] == abi.ST_BAD_INDEX) and not st0[ok].any()
    # BallOnBeam's init space depends on the beam length, so its sampler reads the table
    bob_table = cases.make_case(abi.ENV_BOB, 16, seed=2)['dp']
    init = eng.sample_init(abi.ENV_BOB, n, bob_table, seed=1, dp_index=dp_index).cpu().numpy()
    assert np.isnan(init[:, bad_envs[:5]]).all() and np.isfinite(init[:, ok]).all()


@pytest.mark.parametrize('policy', ['linear', 'fnn'])
@pytest.mark.parametrize('R', [1, 8, 16, 20, 70, 130])
def test_population_mode(eng, orc, policy, R):
    """ ParameterExplorationSampler's layout: env j*R + r runs parameter set j.  For FNN policies a wave runs one policy
    at a time (its weights in LDS), up to 64 of its rollouts: R below, at and above an MFMA tile and above a wave. """
    kind, pop = abi.ENV_QCP_ST, 37
    # the widest layer decides how many weight images fit into LDS, and with that how the tiles of a wave are shared out
    kw = dict(hidden={1: (64, 32), 8: (16, 16), 16: (32, 20), 20: (16, 8), 70: (64, 64, 64), 130: (30,)}[R]) if policy == 'fnn' else {}
    case = cases.make_case(kind, pop*R, seed=13, policy=policy, **kw)
    params = 0.5*np.random.default_rng(2).standard_normal((pop, case['ps'].num_param))
    kwr = dict(policy_group=R, record=True)
    ref = orc.rollout(kind, case['dt'], 40, case['task_c'], case['ps'], params, case['init'], case['dp'], **kwr)
    out = eng.rollout(kind, case['dt'], 40, case['task_c'], case['ps'], params, case['init'], case['dp'], **kwr)
    np.testing.assert_array_equal(out['length'].cpu().numpy(), ref['length'])
    np.testing.assert_array_equal(out['status'].cpu().numpy(), ref['status'])
    for k in ('obs', 'act', 'rew', 'ret', 'final_state'):
        assert cases.rel_err(out[k].cpu().numpy(), ref[k]) <= TOL, k
    # and with wrappers plus exploration noise behind the policy
    import dataclasses
    ps = dataclasses.replace(case['ps'], expl_kind='normal', expl_scale=[0.2])
    wrap = cases.WrapSpec(ops=['act_delay', 'act_norm'], act_delay=1, noise_seed=4)
    kwr = dict(policy_group=R, wrap=wrap)
    ref = orc.rollout(kind, case['dt'], 40, case['task_c'], ps, params, case['init'], case['dp'], **kwr)
    out = eng.rollout(kind, case['dt'], 40, case['task_c'], ps, params, case['init'], case['dp'], **kwr)
    np.testing.assert_array_equal(out['length'].cpu().numpy(), ref['length'])
    assert cases.rel_err(out['ret'].cpu().numpy(), ref['ret']) <= TOL


@pytest.mark.parametrize('kind,pop,R,hidden', [(abi.ENV_OMO, 3000, 4, (16,)), (abi.ENV_QQ, 800, 4, (64, 64)),
                                               (abi.ENV_QCP_ST, 3100, 2, (32, 32)), (abi.ENV_QQ, 800, 40, (64,))],
                         ids=['16-wide', '64-wide', '32-wide', '64-wide-R40'])
def test_large_fnn_population(eng, orc, kind, pop, R, hidden):
    """ Populations large enough for workgroups of several waves (for the 32- and 64-wide nets that is the layout with the
    coarse tanh table) and, in the first case, more tasks than resident groups of tiles: groups take over new
    (policy, chunk of rollouts) tasks and the wave reloads their weight images. """
    case = cases.make_case(kind, pop*R, seed=3, policy='fnn', hidden=hidden, scale=0.3)
    params = 0.3*np.random.default_rng(5).standard_normal((pop, case['ps'].num_param))
    ref = orc.rollout(kind, case['dt'], 30, case['task_c'], case['ps'], params, case['init'], case['dp'], policy_group=R)
    out = eng.rollout(kind, case['dt'], 30, case['task_c'], case['ps'], params, case['init'], case['dp'], policy_group=R)
    np.testing.assert_array_equal(out['length'].cpu().numpy(), ref['length'])
    np.testing.assert_array_equal(out['status'].cpu().numpy(), ref['status'])
    assert cases.rel_err(out['ret'].cpu().numpy(), ref['ret']) <= TOL


@pytest.mark.parametrize('kind,R,hidden', [(abi.ENV_QQ, 8, (64, 64)), (abi.ENV_QQ, 5, (48, 40)), (abi.ENV_QBB, 8, (64, 64)),
                                           (abi.ENV_QBB, 3, (64, 64)), (abi.ENV_OMO, 1, (64, 64)), (abi.ENV_BOB, 7, (40, 64)),
                                           (abi.ENV_QCP_SU, 4, (64, 64)), (abi.ENV_PEND, 8, (33, 17)), (abi.ENV_QQ, 8, (32, 32)),
                                           (abi.ENV_QBB, 6, (24, 32)), (abi.ENV_OMO, 2, (32, 17)), (abi.ENV_QCP_ST, 13, (32, 32))])
def test_populations_of_up_to_8_rollouts_per_policy(eng, orc, kind, R, hidden):
    """ Two-hidden-layer tanh networks wider than 16 with at most 8 rollouts per policy run their layers in column groups
    of 4 rollouts (4x4x4 MFMAs) instead of tiles of 16 (the last case: 13 rollouts, the 32-wide kernel with tiles of 16): every observation size (whole groups of 4 inputs and left-over
    inputs), full and partly filled column groups, padded layer widths, more policies than fit the chip at once, recorded
    trajectories; then a task that is not the simulation's default one, and weights large enough for the clamping tanh. """
    import dataclasses
    pop, T = 2500, 25
    case = cases.make_case(kind, pop*R, seed=21, policy='fnn', hidden=hidden, scale=0.3)
    params = 0.3*np.random.default_rng(6).standard_normal((pop, case['ps'].num_param))
    kwr = dict(policy_group=R, record=True)
    ref = orc.rollout(kind, case['dt'], T, case['task_c'], case['ps'], params, case['init'], case['dp'], **kwr)
    out = eng.rollout(kind, case['dt'], T, case['task_c'], case['ps'], params, case['init'], case['dp'], **kwr)
    np.testing.assert_array_equal(out['length'].cpu().numpy(), ref['length'])
    np.testing.assert_array_equal(out['status'].cpu().numpy(), ref['status'])
    # every rollout was stored, and no wave saw the lanes of a policy group disagree about their common state
    assert np.all(out['length'].cpu().numpy() >= 0) and not np.any(out['status'].cpu().numpy() & abi.ST_INTERNAL)
    for k in ('obs', 'act', 'rew', 'ret', 'final_state'):
        assert cases.rel_err(out[k].cpu().numpy(), ref[k]) <= TOL, k
    # another task (its shape is read at run time), large weights, a NaN in one policy: only its rollouts are affected
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    spec = envdefs.default_task_spec(kind)
    task_c = dataclasses.replace(spec, rew_kind='exp', Q=np.diag(np.linspace(0.5, 2., S)), R=np.diag(np.full(A, 0.3))).to_c(S, A)
    params = np.random.default_rng(7).standard_normal((pop, case['ps'].num_param))
    params[11, 5] = np.nan
    kwr = dict(policy_group=R)
    T = 10
    ref = orc.rollout(kind, case['dt'], T, task_c, case['ps'], params, case['init'], case['dp'], **kwr)
    out = eng.rollout(kind, case['dt'], T, task_c, case['ps'], params, case['init'], case['dp'], **kwr)
    np.testing.assert_array_equal(out['length'].cpu().numpy(), ref['length'])
    np.testing.assert_array_equal(out['status'].cpu().numpy(), ref['status'])
    # every rollout was stored, and no wave saw the lanes of a policy group disagree about their common state
    assert np.all(out['length'].cpu().numpy() >= 0) and not np.any(out['status'].cpu().numpy() & abi.ST_INTERNAL)
    ret, ret_ref = out['ret'].cpu().numpy(), ref['ret']
    np.testing.assert_array_equal(np.isnan(ret), np.isnan(ret_ref))
    ok = ~np.isnan(ret_ref)
    assert cases.rel_err(ret[ok], ret_ref[ok]) <= 1e-7   # saturated bang-bang actions amplify rounding


def test_full_size_cem_generation_is_order_invariant_and_matches_the_oracle_on_a_sample(eng, orc):
    """ A CEM generation at the size of the measurement plan: 4 096 FNN 64x64 policies x 8 rollouts on shared (domain
    parameter, initial state) pairs, QBallBalancer behind ActDelay (per-rollout delay) and ActNorm.  More policies than
    the chip holds at once, so register sets and LDS images are reloaded; which wave, half of a wave or turn runs a policy
    must not influence its results: the generation with its policies permuted returns the permuted results bit for
    bit.  A sample of the policies is checked against the oracle. """
    import torch
    kind, pop, R, T = abi.ENV_QBB, 4096, 8, 120
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    rng = np.random.default_rng(23)
    dp = cases.domain_params(kind, R, rng)
    dp = np.vstack([dp, rng.integers(0, 11, R).astype(float)[None, :]])
    init = np.tile(cases.init_states(kind, dp[:P], rng), (1, pop))
    dp_index = np.tile(np.arange(R), pop).astype(np.int32)
    ps, theta0 = cases.fnn_policy(kind, rng, hidden=(64, 64), scale=0.1)
    params = theta0[None, :] + 0.02*rng.standard_normal((pop, ps.num_param))
    wrap = cases.WrapSpec(ops=['act_delay', 'act_norm'], act_delay_row=P)
    task_c = envdefs.default_task_spec(kind).to_c(S, A)
    dt = cases.DT[kind]
    kw = dict(dp_index=dp_index, policy_group=R, wrap=wrap)
    out = eng.rollout(kind, dt, T, task_c, ps, params, init, dp, **kw)
    perm = rng.permutation(pop)
    out_p = eng.rollout(kind, dt, T, task_c, ps, params[perm], init, dp, **kw)
    for key in ('ret', 'length', 'status'):
        assert torch.equal(out[key].view(pop, R)[torch.as_tensor(perm, device=out[key].device)], out_p[key].view(pop, R)), key
    fs, fs_p = out['final_state'].view(-1, pop, R), out_p['final_state'].view(-1, pop, R)
    assert torch.equal(fs[:, torch.as_tensor(perm, device=fs.device)], fs_p)
    length = out['length'].cpu().numpy()
    assert length.min() < T and length.max() == T        # some balls leave the plate, some rollouts run to the end
    sample = np.sort(rng.choice(pop, 48, replace=False))
    env_idx = (sample[:, None]*R + np.arange(R)[None, :]).ravel()
    ref = orc.rollout(kind, dt, T, task_c, ps, params[sample], init[:, env_idx], dp, dp_index=dp_index[env_idx], policy_group=R,
                      wrap=wrap)
    np.testing.assert_array_equal(length[env_idx], ref['length'])
    np.testing.assert_array_equal(out['status'].cpu().numpy()[env_idx], ref['status'])
    assert cases.rel_err(out['ret'].cpu().numpy()[env_idx], ref['ret']) <= TOL


def test_full_size_launch_is_order_invariant_and_matches_the_oracle_on_a_sample(eng, orc):
    """ The bench's launch size (2 097 152 QQube envs, FNN 64x64, T = 600): which lane, wave or workgroup runs an env
    must not influence its result, so a permuted launch returns the permuted results bit for bit; a sample of the
    envs is checked against the oracle (returns and lengths; the oracle cannot run all of them in test time). """
    import torch
    kind, n, T = abi.ENV_QQ, 1 << 21, 600
    rng = np.random.default_rng(17)
    dp = cases.domain_params(kind, n, rng)
    init = cases.init_states(kind, dp, rng)
    ps, params = cases.fnn_policy(kind, np.random.default_rng(2), hidden=(64, 64), scale=0.25)
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    task_c = envdefs.default_task_spec(kind).to_c(S, A)
    dt = cases.DT[kind]
    dev = eng.device
    dp_d, init_d = torch.as_tensor(dp, device=dev), torch.as_tensor(init, device=dev)
    out = eng.rollout(kind, dt, T, task_c, ps, params, init_d, dp_d)
    perm = torch.randperm(n, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    out_p = eng.rollout(kind, dt, T, task_c, ps, params, init_d[:, perm].contiguous(), dp_d[:, perm].contiguous())
    for key in ('ret', 'length', 'status'):
        assert torch.equal(out[key][perm], out_p[key]), key
    assert torch.equal(out['final_state'][:, perm], out_p['final_state'])
    assert int(out['length'].min()) >= 1 and int(out['length'].max()) <= T
    sample = rng.choice(n, 1500, replace=False)
    ref = orc.rollout(kind, dt, T, task_c, ps, params, init[:, sample], dp[:, sample])
    np.testing.assert_array_equal(out['length'].cpu().numpy()[sample], ref['length'])
    np.testing.assert_array_equal(out['status'].cpu().numpy()[sample], ref['status'])
    # whole rollouts of up to 600 steps: the closed loop amplifies rounding differences, so the bound on the
    # return is looser than the per-step tolerance of the short-horizon tests
    assert cases.rel_err(out['ret'].cpu().numpy()[sample], ref['ret']) <= 1e-6


def test_full_size_qcartpole_swingup_fnn_is_order_invariant_and_matches_the_oracle_on_a_sample(eng, orc):
    """ One GPU's share of the 16 M-environment QCartPoleSwingUp configuration: 2 097 152 envs x T = 1 000 under an FNN
    64x64, dt = 1/500, inputs drawn on the device by the default randomizer.  Permuted launch = permuted results, bit for
    bit; 1 000 envs against the oracle. """
    import torch
    from simurlacra_amd.domain_randomization import get_default_randomizer
    from simurlacra_amd.environments import QCartPoleSwingUpSim
    from bench import reference_fnn_params
    kind, n, T, dt = abi.ENV_QCP_SU, 1 << 21, 1000, 1/500.
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    env = QCartPoleSwingUpSim(dt=dt, max_steps=T)
    dp = eng.randomize(n, get_default_randomizer(env).device_table(envdefs.DP_NAMES[kind], ()), list(env.domain_param_column()), seed=31)
    init = eng.sample_init(kind, n, dp, seed=31)
    task_c = env.task_spec().to_c(S, A)
    ps = cases.PolicySpec('fnn', O, A, hidden=[64, 64], hidden_nonlin=['tanh', 'tanh'])
    params = reference_fnn_params([O, 64, 64, A], 0)
    out = eng.rollout(kind, dt, T, task_c, ps, params, init, dp)
    perm = torch.randperm(n, device=eng.device, generator=torch.Generator(device=eng.device).manual_seed(6))
    out_p = eng.rollout(kind, dt, T, task_c, ps, params, init[:, perm].contiguous(), dp[:, perm].contiguous())
    for key in ('ret', 'length', 'status'):
        assert torch.equal(out[key][perm], out_p[key]), key
    assert torch.equal(out['final_state'][:, perm], out_p['final_state'])
    assert int(out['length'].min()) >= 1 and int(out['length'].max()) == T
    sample = np.random.default_rng(3).choice(n, 1000, replace=False)
    idx = torch.as_tensor(sample, device=eng.device)
    ref = orc.rollout(kind, dt, T, task_c, ps, params.numpy(), init[:, idx].cpu().numpy(), dp[:, idx].cpu().numpy())
    np.testing.assert_array_equal(out['length'].cpu().numpy()[sample], ref['length'])
    np.testing.assert_array_equal(out['status'].cpu().numpy()[sample], ref['status'])
    assert cases.rel_err(out['ret'].cpu().numpy()[sample], ref['ret']) <= 1e-6   # whole rollouts of 1 000 steps


def test_full_size_qqube_sweep_over_parameter_sets_is_order_invariant_and_matches_the_oracle_on_a_sample(eng, orc):
    """ The sweep layout (SPOTA / BayRn inner loops): 2 097 152 QQube rollouts that read their domain parameters through
    `dp_index` from a table of 8 parameter sets (262 144 rollouts per set), FNN 64x64, T = 600.  The same launch with
    rollouts AND table columns permuted returns the permuted results bit for bit; it equals the launch with the table
    expanded to one column per rollout; 1 000 rollouts against the oracle (which reads the table the same way). """
    import torch
    kind, n_sets, R, T, dt = abi.ENV_QQ, 8, 1 << 18, 600, 1/500.
    n = n_sets*R
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    rng = np.random.default_rng(41)
    table = cases.domain_params(kind, n_sets, rng, spread=0.15)
    dp_index = np.repeat(np.arange(n_sets, dtype=np.int32), R)
    init = cases.init_states(kind, np.zeros((1, n)), rng)
    ps, params = cases.fnn_policy(kind, np.random.default_rng(2), hidden=(64, 64), scale=0.25)
    task_c = envdefs.default_task_spec(kind).to_c(S, A)
    dev = eng.device
    table_d, init_d, idx_d = eng.f64(table), eng.f64(init), eng.i32(dp_index)
    out = eng.rollout(kind, dt, T, task_c, ps, params, init_d, table_d, dp_index=idx_d)
    perm = torch.randperm(n, device=dev, generator=torch.Generator(device=dev).manual_seed(8))
    cols = torch.tensor([5, 2, 7, 0, 3, 6, 1, 4], device=dev)                      # new column j holds old column cols[j]
    where = torch.empty_like(cols); where[cols] = torch.arange(n_sets, device=dev)  # old column c now sits at where[c]
    out_p = eng.rollout(kind, dt, T, task_c, ps, params, init_d[:, perm].contiguous(), table_d[:, cols].contiguous(),
                        dp_index=where[idx_d.long()[perm]].to(torch.int32))
    full = eng.rollout(kind, dt, T, task_c, ps, params, init_d, table_d[:, idx_d.long()].contiguous())
    for key in ('ret', 'length', 'status'):
        assert torch.equal(out[key][perm], out_p[key]), key
        assert torch.equal(out[key], full[key]), key
    # every set saw different physics: the sets' mean returns differ
    assert len(set(np.round(out['ret'].view(n_sets, R).mean(dim=1).cpu().numpy(), 6))) == n_sets
    sample = rng.choice(n, 1000, replace=False)
    ref = orc.rollout(kind, dt, T, task_c, ps, params, init[:, sample], table, dp_index=dp_index[sample])
    np.testing.assert_array_equal(out['length'].cpu().numpy()[sample], ref['length'])
    np.testing.assert_array_equal(out['status'].cpu().numpy()[sample], ref['status'])
    assert cases.rel_err(out['ret'].cpu().numpy()[sample], ref['ret']) <= 1e-6


@pytest.mark.parametrize('kind', cases.KINDS, ids=[cases.KIND_NAMES[k] for k in cases.KINDS])
def test_return_distributions_over_the_whole_horizon_match(eng, orc, kind):
    """ Closed loops amplify rounding, so over hundreds of steps single rollouts of device and oracle may part; what the
    algorithms consume is the distribution of returns.  10 000 rollouts per simulation over its whole horizon under a
    64x64 tanh network with gains large enough for episodes to fail: the two samples of returns (same initial states
    and parameters) must be indistinguishable -- two-sample Kolmogorov-Smirnov statistic, mean, quantiles -- and all but
    a handful of rollouts must agree in length and, to 1e-6, in return. """
    from scipy import stats
    n, T = 10000, 600 if kind in (abi.ENV_QQ, abi.ENV_QQ_ST) else 500
    case = cases.make_case(kind, n, seed=900 + kind, policy='fnn', hidden=(64, 64), scale=0.25)
    args = (kind, case['dt'], T, case['task_c'], case['ps'], case['params'], case['init'], case['dp'])
    ref = orc.rollout(*args)
    out = eng.rollout(*args)
    ret, length = out['ret'].cpu().numpy(), out['length'].cpu().numpy()
    same_len = np.mean(length == ref['length'])
    close = np.mean(np.abs(ret - ref['ret']) <= 1e-6*np.maximum(1., np.abs(ref['ret'])))
    spread = max(np.std(ref['ret']), 1e-12)
    ks = stats.ks_2samp(ret, ref['ret'], method='asymp').statistic
    qs = [5, 25, 50, 75, 95]
    dq = np.max(np.abs(np.percentile(ret, qs) - np.percentile(ref['ret'], qs)))/spread
    dmean = abs(ret.mean() - ref['ret'].mean())/spread
    msg = f'{cases.KIND_NAMES[kind]}: same length {same_len:.5f}, returns within 1e-6 {close:.5f}, KS {ks:.2e}, ' \
          f'quantiles {dq:.2e} and mean {dmean:.2e} of the std; mean length {length.mean():.1f}'
    print(msg)
    assert same_len >= 0.999 and close >= 0.995, msg
    assert ks <= 2e-3 and dq <= 2e-3 and dmean <= 1e-4, msg


def test_large_launches_of_deeper_nets(eng, orc):
    """ Launch sizes at which the bare FNN kernel runs 12 waves per workgroup, with weight images of growing size (the
    staging areas of the extra waves must not cost the network its tanh table or its place in LDS): returns of a
    sample against the oracle. """
    import torch
    kind, n, T = abi.ENV_QQ, 420000, 12
    rng = np.random.default_rng(3)
    dp = cases.domain_params(kind, n, rng)
    init = cases.init_states(kind, dp, rng)
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    task_c = envdefs.default_task_spec(kind).to_c(S, A)
    sample = rng.choice(n, 500, replace=False)
    for hidden in ((64, 64), (64, 64, 64), (64, 64, 64, 64)):
        ps, params = cases.fnn_policy(kind, np.random.default_rng(len(hidden)), hidden=hidden, scale=0.2)
        out = eng.rollout(kind, cases.DT[kind], T, task_c, ps, params, torch.as_tensor(init, device=eng.device),
                          torch.as_tensor(dp, device=eng.device))
        ref = orc.rollout(kind, cases.DT[kind], T, task_c, ps, params, init[:, sample], dp[:, sample])
        np.testing.assert_array_equal(out['length'].cpu().numpy()[sample], ref['length'])
        assert cases.rel_err(out['ret'].cpu().numpy()[sample], ref['ret']) <= TOL


def test_ragged_and_empty_batches(eng, orc):
    """ N not a multiple of the wave/block size, N = 1 and N = 0 """
    kind = abi.ENV_QQ
    for n in (1, 63, 65, 513):
        case = cases.make_case(kind, n, seed=n, policy='fnn', hidden=(64, 64))
        _compare_rollouts(eng, orc, case, T=30)
    case = cases.make_case(kind, 4, seed=1, policy='linear')
    out = eng.rollout(kind, case['dt'], 10, case['task_c'], case['ps'], case['params'], case['init'][:, :0], case['dp'][:, :1])
    assert out['ret'].numel() == 0


@pytest.mark.parametrize('n,hidden', [(15, (64, 64)), (17, (64, 64)), (4095, (64, 64)), (4097, (64, 64)), (8191, (64, 64)),
                                      (8193, (64, 64)), (16385, (64, 64)), (32769, (64, 64)), (65537, (64, 64)),
                                      (98305, (64, 64)), (131073, (64, 64)), (15, (64,)), (4097, (64, 32, 16)), (8191, (40, 64, 8)),
                                      (8191, (32, 32)), (8193, (32, 24)), (37, (17, 32))])
def test_small_and_medium_fnn_launches_at_the_edges_of_their_layouts(eng, orc, n, hidden):
    """ FNN launches below two full waves per SIMD are spread over more waves with fewer MFMA tiles each (1-4 tiles per wave,
    1, 4 or 8 waves per workgroup); a 32- or 64-wide two-layer tanh network on up to 8 192 environments runs as a population
    of copies of the policy with 8 (or 4) rollouts each.  Batch sizes next to every switch of those layouts, bare and behind a
    wrapper chain with an action delay (whose ring buffers are indexed by the resident lane), against the oracle. """
    kind, T = abi.ENV_QCP_ST, 12
    case = cases.make_case(kind, n, seed=n, policy='fnn', hidden=hidden, scale=0.3)
    args = (kind, case['dt'], T, case['task_c'], case['ps'], case['params'], case['init'], case['dp'])
    for wrap in (None, cases.WrapSpec(ops=['act_delay', 'act_norm'], act_delay=2)):
        ref = orc.rollout(*args, wrap=wrap)
        out = eng.rollout(*args, wrap=wrap)
        np.testing.assert_array_equal(out['length'].cpu().numpy(), ref['length'])
        np.testing.assert_array_equal(out['status'].cpu().numpy(), ref['status'])
        assert cases.rel_err(out['ret'].cpu().numpy(), ref['ret']) <= TOL
        assert cases.rel_err(out['final_state'].cpu().numpy(), ref['final_state']) <= TOL


def test_one_engine_from_two_threads(eng, orc):
    """ Rollouts issued from two host threads through one engine (one context): the calls are serialised on the host and
    ordered on the device, so each gets the results of a call made alone. """
    import threading
    kind, T = abi.ENV_QQ, 40
    jobs = [cases.make_case(kind, 3000 + 500*j, seed=50 + j, policy='fnn' if j % 2 else 'linear', scale=0.3) for j in range(4)]
    alone = [eng.rollout(kind, c['dt'], T, c['task_c'], c['ps'], c['params'], c['init'], c['dp']) for c in jobs]
    got = [None]*len(jobs)

    def work(j0):
        import torch
        stream = torch.cuda.Stream(device=eng.device)
        with torch.cuda.stream(stream):
            for rep in range(5):
                for j in range(j0, len(jobs), 2):
                    c = jobs[j]
                    got[j] = eng.rollout(kind, c['dt'], T, c['task_c'], c['ps'], c['params'], c['init'], c['dp'])
            stream.synchronize()
    threads = [threading.Thread(target=work, args=(j0,)) for j0 in (0, 1)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    import torch
    for a, g in zip(alone, got):
        for key in ('ret', 'length', 'status', 'final_state'):
            assert torch.equal(a[key], g[key]), key


def test_stop_on_done_false_and_long_horizon_failures(eng, orc):
    """ rollout(stop_on_done=False): the env keeps stepping after done, the final reward is given once. """
    kind = abi.ENV_OMO
    case = cases.make_case(kind, 128, seed=3, policy='linear', scale=3.0)
    _compare_rollouts(eng, orc, case, T=120, flags=0)
    _compare_rollouts(eng, orc, case, T=120, flags=abi.F_STOP_ON_DONE)


def test_nan_actions_are_flagged_not_fatal(eng, orc):
    kind = abi.ENV_BOB
    case = cases.make_case(kind, 64, seed=9, policy='linear')
    params = case['params'].copy()
    params[0] = np.nan
    ref = orc.rollout(kind, case['dt'], 20, case['task_c'], case['ps'], params, case['init'], case['dp'])
    out = eng.rollout(kind, case['dt'], 20, case['task_c'], case['ps'], params, case['init'], case['dp'])
    np.testing.assert_array_equal(out['status'].cpu().numpy(), ref['status'])
    np.testing.assert_array_equal(out['length'].cpu().numpy(), ref['length'])
    assert np.all(out['status'].cpu().numpy() & abi.ST_NAN)


@pytest.mark.parametrize('kind', cases.KINDS, ids=lambda k: cases.KIND_NAMES[k])
@pytest.mark.parametrize('policy', ['identity', 'features', 'fnn', 'fnn-padded'])
def test_nan_and_infinite_initial_states(eng, orc, kind, policy):
    """ rollout() tests the observation and the action for NaN (sampling/rollout.py:172-175, 207-210).  The kernels for a
    LinearPolicy on the plain observations leave the first test to the second (a NaN observation makes the action a
    NaN): status, length and the recorded rows must come out as with both tests, for a NaN and for an infinite state
    component (whose sine is a NaN, or which fails the bounds at the first step), next to rollouts that run.  What an
    infinite component turns into follows IEEE arithmetic in the reference -- x / inf = 0, inf / x = inf, 0 * inf = NaN in
    its matrix products, tanh(inf) = 1 -- and decides the NaN bit of the status word: every recorded value has the
    oracle's NaN and its infinities. """
    kw = {'identity': dict(feats=('identity',)), 'features': dict(), 'fnn': dict(hidden=(64, 64)), 'fnn-padded': dict(hidden=(20, 7))}[policy]
    case = cases.make_case(kind, 192, seed=40 + kind, policy='fnn' if policy.startswith('fnn') else 'linear', **kw)
    init = case['init'].copy()
    S = init.shape[0]
    for j in range(0, 192, 3):
        init[(j // 3) % S, j] = [np.nan, np.inf, -np.inf][(j // 3) % 3]
    args = (kind, case['dt'], 25, case['task_c'], case['ps'], case['params'], init, case['dp'])
    for flags in (abi.F_STOP_ON_DONE, 0):
        ref = orc.rollout(*args, record=True, flags=flags)
        got = {k: v.cpu().numpy() for k, v in eng.rollout(*args, record=True, flags=flags).items()}
        np.testing.assert_array_equal(got['length'], ref['length'])
        np.testing.assert_array_equal(got['status'], ref['status'])
        assert (ref['status'] & abi.ST_NAN).any() and (ref['length'] > 5).any()
        for k in ('act', 'rew', 'ret', 'obs', 'final_state'):
            assert cases.rel_err(got[k], ref[k]) <= TOL, k          # (includes the NaN pattern)
            inf = np.isinf(ref[k])
            np.testing.assert_array_equal(np.isinf(got[k]), inf, err_msg=k)
            np.testing.assert_array_equal(got[k][inf], ref[k][inf], err_msg=k)


def _ulp_err(got, ref):
    return np.abs(got - ref)/np.spacing(np.abs(ref))


def test_device_math_accuracy(eng):
    """ The step loop's own fp64 exp/tanh/sigmoid/sin/cos/fmod against libm (NumPy): a few ulp on the ranges the
    simulations produce, exact for fmod, library behaviour outside the fast range. """
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-700, 700, 20000), rng.uniform(-2, 2, 20000), [0., -0., 1e-300, -745.2, 709.7]])
    assert _ulp_err(eng.eval_math('exp', x).cpu().numpy(), np.exp(x)).max() <= 2
    x = np.concatenate([rng.uniform(-25, 25, 20000), rng.uniform(-1, 1, 20000), 10.**rng.uniform(-12, 0, 5000), [0., 40., -40.]])
    x = np.concatenate([x, np.arange(0, 5121)/256., np.arange(0, 5121)/256. + 1/512., [20., 20.000001, 1e300, -1e300]])
    for fn in ('tanh', 'tanh_table_fine', 'tanh_table_coarse'):  # exponential form and the two table forms of the FNN kernel
        got, ref = eng.eval_math(fn, x).cpu().numpy(), np.tanh(x)
        assert _ulp_err(got[ref != 0], ref[ref != 0]).max() <= 4 and np.all(got[ref == 0] == 0), fn
        assert np.all(np.signbit(got) == np.signbit(x)), fn
    x = np.concatenate([rng.uniform(-40, 40, 20000), [0., 800., -800.]])
    got, ref = eng.eval_math('sigmoid', x).cpu().numpy(), 1/(1 + np.exp(-x))
    assert np.abs(got - ref).max() <= 3e-16 and _ulp_err(got[np.abs(x) < 30], ref[np.abs(x) < 30]).max() <= 4
    x = np.concatenate([rng.uniform(-100, 100, 40000), rng.uniform(-1e5, 1e5, 10000), [0., 1e7, -3e9, np.pi, np.pi/2]])
    for fn, f in (('sin', np.sin), ('cos', np.cos)):
        got, ref = eng.eval_math(fn, x).cpu().numpy(), f(x)
        assert np.abs(got - ref).max() <= 2.5e-16, fn
    xs = np.concatenate([rng.uniform(-50, 50, 20000), [-2., -4., 4., 2*np.pi, -6*np.pi, 7., 0., 1e300]])
    ms = np.concatenate([rng.choice([2*np.pi, 2., 3., 0.37], 20000), [2., 3., 2., 2*np.pi, 2*np.pi, 7., 2., 2*np.pi]])
    got, ref = eng.eval_math('fmod', xs, ms).cpu().numpy(), np.fmod(xs, ms)
    np.testing.assert_array_equal(got, ref)
    np.testing.assert_array_equal(np.signbit(got), np.signbit(ref))
    nan_in = np.array([np.nan, np.inf, -np.inf])
    for fn in ('tanh', 'tanh_table_fine', 'tanh_table_coarse'):
        got = eng.eval_math(fn, nan_in).cpu().numpy()
        assert np.isnan(got[0]) and got[1] == 1. and got[2] == -1., fn
    assert np.isnan(eng.eval_math('exp', nan_in).cpu().numpy()[0]) and np.all(np.isnan(eng.eval_math('sin', nan_in).cpu().numpy()))
    got = eng.eval_math('sigmoid', nan_in).cpu().numpy()
    assert np.isnan(got[0]) and got[1] == 1. and got[2] == 0.
    got = eng.eval_math('exp', nan_in).cpu().numpy()
    assert got[1] == np.inf and got[2] == 0.
    # log: the swing-up reward's argument range, the Box-Muller draws' (0, 1], the whole exponent range, and the
    # arguments the fast path hands to the library
    x = np.concatenate([rng.uniform(0.1, 10.2, 20000), 1. - rng.random(20000), 2.**rng.uniform(-1000, 1000, 20000),
                        1. + rng.uniform(-1e-3, 1e-3, 5000), [1., 0.5, 2., np.sqrt(0.5), np.sqrt(2.), 2.**-53, 1e-310, 1e308]])
    got, ref = eng.eval_math('log', x).cpu().numpy(), np.log(x)
    nz = ref != 0
    assert _ulp_err(got[nz], ref[nz]).max() <= 1.5 and np.all(got[~nz] == 0)
    with np.errstate(all='ignore'):
        odd = np.array([0., -1., np.inf, np.nan])
        got = eng.eval_math('log', odd).cpu().numpy()
    assert got[0] == -np.inf and np.isnan(got[1]) and got[2] == np.inf and np.isnan(got[3])


@pytest.mark.parametrize('kind', [abi.ENV_QQ, abi.ENV_QBB], ids=['qq', 'qbb'])
def test_a_wrapper_list_without_noise_does_not_pay_for_the_noise_draws(eng, kind):
    """ The wrapped kernels carry the Philox + Box-Muller draws of GaussianObsNoiseWrapper / GaussianActNoiseWrapper; a
    launch whose wrapper list holds neither must not execute them.  (They once ran on every step of every wrapped launch,
    hoisted in front of the loop over the wrapper list: four to five times the instructions of the bare kernel.)  262 144
    full-length LinearPolicy rollouts behind ActNormWrapper, timed with HIP events against the bare launch. """
    import torch
    from simurlacra_amd.specs import WrapSpec
    n = 1 << 18
    case = cases.make_case(kind, n, seed=4, policy='linear', feats=['identity'], scale=0.0)
    args = (kind, case['dt'], 300, case['task_c'], case['ps'], torch.as_tensor(case['params'], device=eng.device),
            torch.as_tensor(case['init'], device=eng.device), torch.as_tensor(case['dp'], device=eng.device))

    def timed(**kw):
        best = np.inf
        for _ in range(3):
            out = eng.rollout(*args, final_state=False, **kw)
            best = min(best, eng.last_rollout_ms())
        return best, out

    t_bare, bare = timed()
    t_norm, norm = timed(wrap=WrapSpec(ops=['act_norm']))
    # zero weights: the normalised action 0 is the middle of the symmetric action range, the rollouts are the same
    assert torch.equal(bare['length'], norm['length'])
    assert t_norm < 1.9*t_bare, (t_bare, t_norm)


@pytest.mark.parametrize('kind,min_gbs', [(abi.ENV_OMO, 3000.), (abi.ENV_BOB, 1600.)], ids=['omo', 'bob'])
def test_trajectory_writeback_at_four_times_the_resident_lanes(eng, orc, kind, min_gbs):
    """ 2^21 environments with full trajectories (T = 500): OneMassOscillator runs every episode to the end (34 GB of
    trajectory), BallOnBeam ends after 69 steps on average, so every lane is refilled seven times and the wave's lanes
    takes part in about seven groups of 16 consecutive environments that run in lock step (whole 128-byte lines per store, the
    rows of an environment that ended before its group padded with NaN).  A sample of environments is compared with the
    oracle step by step; for ALL of them the recorded rewards must add up to the returned return and the rows past an
    episode's end must hold NaN or what the caller passed; the writeback rate must stay within reach of what profiles/
    records. """
    import torch
    n, T = 1 << 21, 500
    S, O, A, P, I = abi.ENV_DIMS[kind]
    nom = envdefs.nominal_domain_param(kind)
    distrs = [dict(kind=abi.DISTR_NORMAL, row=r, a=v, b=abs(v)/5, clip_lo=1e-3*abs(v)) for r, v in enumerate(nom.values()) if v != 0]
    dp = eng.randomize(n, distrs, list(nom.values()), seed=1)
    init = eng.sample_init(kind, n, dp, seed=1)
    ps, params = cases.linear_policy(kind, np.random.default_rng(0), feats=('identity',), scale=0.1)
    task_c = envdefs.default_task_spec(kind).to_c(S, A)
    dev = eng.device
    marker = -12345.0
    out = dict(ret=torch.empty(n, dtype=torch.float64, device=dev), length=torch.empty(n, dtype=torch.int32, device=dev),
               status=torch.empty(n, dtype=torch.int32, device=dev),
               obs=torch.full((T + 1, O, n), marker, dtype=torch.float64, device=dev),
               act=torch.full((T, A, n), marker, dtype=torch.float64, device=dev),
               rew=torch.full((T, n), marker, dtype=torch.float64, device=dev))
    eng.rollout(kind, cases.DT[kind], T, task_c, ps, params, init, dp, final_state=False, out=out)
    eng.rollout(kind, cases.DT[kind], T, task_c, ps, params, init, dp, final_state=False, out=out)
    ms = eng.last_rollout_ms()
    length = out['length'].long()
    steps = int(length.sum().item())
    # every environment: rows below its length written, rows above untouched, rewards add up to the return
    k = torch.arange(T, device=dev).unsqueeze(1)
    live = k < length.unsqueeze(0)
    untouched_or_nan = lambda t: (t == marker) | t.isnan()
    assert bool((untouched_or_nan(out['rew']) == ~live).all())
    assert bool((untouched_or_nan(out['obs'][:, 0, :]) == ~(torch.arange(T + 1, device=dev).unsqueeze(1) <= length.unsqueeze(0))).all())
    assert bool((untouched_or_nan(out['act'][:, 0, :]) == ~live).all())
    total = torch.where(live, out['rew'], torch.zeros((), dtype=torch.float64, device=dev)).sum(dim=0)
    assert float(((total - out['ret']).abs()/out['ret'].abs().clamp(min=1.)).max()) <= 1e-9
    # a sample of environments against the oracle, step by step
    idx = torch.cat([torch.arange(0, 64), torch.arange(n - 64, n), torch.randint(0, n, (384,), generator=torch.Generator().manual_seed(0))]).to(dev)
    ref = orc.rollout(kind, cases.DT[kind], T, task_c, ps, params, init[:, idx].cpu().numpy(), dp[:, idx].cpu().numpy(), record=True)
    np.testing.assert_array_equal(out['length'][idx].cpu().numpy(), ref['length'])
    np.testing.assert_array_equal(out['status'][idx].cpu().numpy(), ref['status'])
    for key in ('obs', 'act', 'rew'):
        got = out[key][..., idx].cpu().numpy()
        got[got == marker] = np.nan   # (padded rows hold NaN already)
        assert cases.rel_err(got, ref[key]) <= TOL, key
    assert cases.rel_err(out['ret'][idx].cpu().numpy(), ref['ret']) <= TOL
    gbs = (steps*(O + A + 1) + n*O)*8/ms/1e6
    print(f'{cases.KIND_NAMES[kind]}: {steps/n:.1f} steps per episode, {ms:.2f} ms, {gbs:.0f} GB/s of trajectory bytes')
    assert gbs >= min_gbs


def test_maximum_sizes_and_one_beyond(eng, orc):
    """ The limits of the descriptor (PSH_MAX_*): a LinearPolicy with 8 feature functions and 512 feature values, an
    FNNPolicy of 4 hidden layers of 64 units, an ActDelayWrapper of 32 steps (fixed, and per environment from the domain
    parameter table) run like any other launch; one beyond each limit is refused with an error, on both sides of the ABI. """
    from simurlacra_amd.specs import FeatSpec, PolicySpec, WrapSpec
    from simurlacra_amd.engine import EngineError
    kind, n, T = abi.ENV_QBB, 500, 40
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    rng = np.random.default_rng(2)
    base = cases.make_case(kind, n, seed=8, policy='linear')

    def check(ps, params, wrap=None, dp=None, tol=TOL):
        args = (kind, base['dt'], T, base['task_c'], ps, params, base['init'], base['dp'] if dp is None else dp)
        ref = orc.rollout(*args, record=True, wrap=wrap)
        out = {k: v.cpu().numpy() for k, v in eng.rollout(*args, record=True, wrap=wrap).items()}
        np.testing.assert_array_equal(out['length'], ref['length'])
        np.testing.assert_array_equal(out['status'], ref['status'])
        for k in ('obs', 'act', 'rew', 'ret'):
            assert cases.rel_err(out[k], ref[k]) <= tol, k

    # 8 feature functions, 1 + 8 + 8 + 8 + 8 + 1 + 382 + 96 = 512 feature values
    D, K = 382, 12
    centers = np.stack([np.linspace(-1.5, 1.5, K) for _ in range(O)], axis=1)
    feats = ['const', 'identity', 'squared', 'sin', 'cos', FeatSpec('mult', idcs=[0, 3, 5]),
             FeatSpec('rff', scale=np.sqrt(2./D), freq=rng.standard_normal((D, O)), shift=2*np.pi*rng.random(D)),
             FeatSpec('rbf', centers=centers, scale=np.full(O, 3.), state_wise=True)]
    ps = PolicySpec('linear', O, A, feats=feats)
    assert len(feats) == abi.PSH_MAX_FEATS and ps.num_feat == abi.PSH_MAX_LIN_FEAT
    check(ps, 0.05*rng.standard_normal(ps.num_param))
    # the widest and deepest network
    ps = PolicySpec('fnn', O, A, hidden=[abi.PSH_MAX_WIDTH]*abi.PSH_MAX_HIDDEN, hidden_nonlin=['tanh']*abi.PSH_MAX_HIDDEN)
    check(ps, 0.15*rng.standard_normal(ps.num_param))
    # the longest delay, fixed and per environment (0..32, as the extra row of the domain parameter table)
    lin = PolicySpec('linear', O, A, feats=['identity'])
    w = 0.3*rng.standard_normal(lin.num_param)
    check(lin, w, wrap=WrapSpec(ops=['act_delay'], act_delay=abi.PSH_MAX_DELAY))
    dp = np.vstack([base['dp'], rng.integers(0, abi.PSH_MAX_DELAY + 1, n).astype(float)[None, :]])
    check(lin, w, wrap=WrapSpec(ops=['act_norm', 'act_delay'], act_delay_row=P), dp=dp)
    # one beyond
    for make in (lambda: PolicySpec('fnn', O, A, hidden=[16]*(abi.PSH_MAX_HIDDEN + 1), hidden_nonlin=['tanh']*(abi.PSH_MAX_HIDDEN + 1)),
                 lambda: PolicySpec('fnn', O, A, hidden=[abi.PSH_MAX_WIDTH + 1], hidden_nonlin=['tanh']),
                 lambda: PolicySpec('linear', O, A, feats=feats + ['abs']),
                 lambda: PolicySpec('linear', O, A, feats=feats[:6] + [FeatSpec('rff', scale=1., freq=rng.standard_normal((D + 97, O)), shift=np.zeros(D + 97))])):
        bad = make()
        with pytest.raises((ValueError, EngineError)):
            eng.rollout(kind, base['dt'], T, base['task_c'], bad, np.zeros(bad.num_param), base['init'], base['dp'])
    with pytest.raises((ValueError, EngineError)):
        eng.rollout(kind, base['dt'], T, base['task_c'], lin, w, base['init'], base['dp'], wrap=WrapSpec(ops=['act_delay'], act_delay=abi.PSH_MAX_DELAY + 1))
    # (the library itself refuses what a caller who fills the descriptor by hand gets wrong: a width beyond the limit)
    good = PolicySpec('fnn', O, A, hidden=[64], hidden_nonlin=['tanh'])
    class Wide(PolicySpec):
        def to_c(self, *a, **kw):
            p = PolicySpec.to_c(self, *a, **kw)
            p.hidden[0] = abi.PSH_MAX_WIDTH + 1
            return p
    wide = Wide('fnn', O, A, hidden=[64], hidden_nonlin=['tanh'])
    with pytest.raises(EngineError):
        eng.rollout(kind, base['dt'], T, base['task_c'], wide, np.zeros(good.num_param + 200), base['init'], base['dp'])


def test_randomized_launches(eng):
    """ A slice of tools/fuzz_launches.py in the suite: 400 launches whose simulation, sizes, policy, assignment of policies
    and domain parameters, flags, wrapper chain, exploration, task and entry point are drawn at random, with NaN weights,
    NaN and infinite initial states among them, against the C oracle -- lengths and status words exact, values to 1e-9
    or within the oracle's own response to one-ulp changes of its inputs. """
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'fuzz_launches.py')
    spec = importlib.util.spec_from_file_location('fuzz_launches', path)
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    lines = []
    bad, refused = fuzz.main(range(700000, 700400), eng, inf=True, wild=False, gentle=True,
                            log=lambda *a, **k: lines.append(' '.join(map(str, a))))
    # (gentle: gains and horizons that keep every closed loop from amplifying rounding differences, so the outcome does not
    # hinge on the tool's judgement of chaotic cases; the tool itself runs them)
    assert bad == 0, '\n'.join(l for l in lines if 'MISMATCH' in l)
    assert sum('(explained)' in l for l in lines) <= 3
    assert refused < 80    # (an FNNPolicy with an index per environment, ActDelayWrapper through the static entry point)


def test_randomized_steps_and_sampling(eng):
    """ A slice of tools/fuzz_steps.py in the suite: psh_reset + psh_step under random actions (outside the bounds, zeros,
    NaN, past `done`), psh_randomize with random lists of distributions, psh_sample_init, psh_topk with ties and NaNs and
    psh_segment_mean, 200 trials each, against the C oracle and NumPy. """
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'fuzz_steps.py')
    spec = importlib.util.spec_from_file_location('fuzz_steps', path)
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    lines = []
    assert fuzz.main(range(800000, 800200), eng, out=lines.append) == 0, '\n'.join(lines)
