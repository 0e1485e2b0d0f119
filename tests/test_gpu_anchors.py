"""
GPU parity tests that do not lean on closed-loop agreement:
  * the kernels against hand-computed one-step numbers (tests/anchors.py) and the inverse kinematics against
    torch.optim.SGD + autograd;
  * teacher forcing: the ORACLE's recorded actions go through psh_step, so device and oracle integrate identical
    initial states, domain parameters and actions over the whole horizon (600 steps of the benchmark workload);
  * the policy alone: the device's FNN / LinearPolicy forward on given observations against the C oracle and against
    torch.nn.Linear in fp64.
Tolerances: 1e-9 relative per step for the dynamics (the stated bar), 1e-12 for a single policy forward.
"""
import numpy as np
import pytest
import torch

from simurlacra_amd import _abi as abi, envdefs
from simurlacra_amd.specs import PolicySpec
from tests import anchors, cases
from tests.test_anchors import STEP_ANCHORS, IDS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from simurlacra_amd.engine import get_engine
    return get_engine(0)


@pytest.fixture(scope='module')
def orc():
    from oracle import oracle_lib
    return oracle_lib


# ------------------------------------------------------------------------------------------- hand-computed anchors
@pytest.mark.parametrize('name,kind,make,opt', STEP_ANCHORS, ids=IDS)
def test_device_one_step_matches_hand_computed(eng, name, kind, make, opt):
    a = make()
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    dp = np.array(a['dp'])[:, None]
    task_c = envdefs.default_task_spec(kind).to_c(S, A)
    flags = abi.F_SIMPLIFIED_DYN if opt.get('simplified') else 0
    mem = eng.reset(kind, a['dt'], 100, task_c, np.array(a['state'])[:, None], dp, flags=flags)
    if 'plate' in a:
        mem['state'][S:S + 2, 0] = torch.tensor(a['plate'], dtype=torch.float64, device=eng.device)
    act = eng.f64(np.array(a['act'])[:, None])
    obs, rew, done, st = eng.step(kind, a['dt'], 100, task_c, mem, act, dp, flags=flags)
    state = mem['state'].cpu().numpy()
    np.testing.assert_allclose(state[:S, 0], a['next'], rtol=1e-12, atol=0)
    np.testing.assert_allclose(obs.cpu().numpy()[:, 0], a['obs'], rtol=1e-12, atol=0)
    assert float(rew[0]) == pytest.approx(a['rew'], rel=1e-12)
    assert int(done[0]) == 0 and int(st[0]) == 0
    if 'plate_next' in a:
        np.testing.assert_allclose(state[S:S + 2, 0], a['plate_next'], rtol=1e-12, atol=0)
        np.testing.assert_array_equal(act.cpu().numpy()[:, 0], a['act'] if opt.get('simplified') else a['act_after'])


def test_device_ik_matches_torch_sgd_with_autograd(eng):
    pts = [(-1.4, 0.0254, 0.275), (-0.5, 0.03, 0.25), (0., 0.0254, 0.275), (0.3, 0.02, 0.3), (1.45, 0.0254, 0.275)]
    th, r, lp = (np.array(v) for v in zip(*pts))
    got = eng.qbb_ik(th, r, lp).cpu().numpy()
    want = [anchors.qbb_ik_torch(*p) for p in pts]
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)


# ------------------------------------------------------------------------------------------------ teacher forcing
def _teacher_forced(eng, orc, case, T, flags=0):
    """ Oracle rollout with its own policy; its recorded actions then drive psh_step and the oracle's own step function
    from the same initial states and parameters.  (The oracle is stepped again because QBallBalancerSim records the
    action AFTER its dead zone, pysim/quanser_ball_balancer.py:212-215: the replay reproduces the rollout's states bit
    for bit, but its rewards see the recorded instead of the raw action.)  Returns the per-step relative deviation of
    observations and rewards, maximised over the environments still running, and the lengths. """
    kind, dt = case['kind'], case['dt']
    ref = orc.rollout(kind, dt, T, case['task_c'], case['ps'], case['params'], case['init'], case['dp'], record=True,
                      flags=abi.F_STOP_ON_DONE | flags)
    N = case['init'].shape[1]
    length = ref['length']
    steps = int(length.max())
    dp = eng.f64(case['dp'])
    mem = eng.reset(kind, dt, T, case['task_c'], case['init'], dp, flags=flags)
    mem_ref = orc.reset(kind, dt, T, case['task_c'], case['init'], case['dp'], flags=flags)
    acts = np.nan_to_num(ref['act'], nan=0.)             # lanes past their end receive zeros and are ignored
    acts_dev = eng.f64(acts)
    obs_all = torch.empty((steps,) + tuple(mem['obs'].shape), dtype=torch.float64, device=eng.device)
    rew_all = torch.empty((steps, N), dtype=torch.float64, device=eng.device)
    done_all = torch.empty((steps, N), dtype=torch.int32, device=eng.device)
    for k in range(steps):
        o, r, d, st = eng.step(kind, dt, T, case['task_c'], mem, acts_dev[k].contiguous(), dp, flags=flags)
        obs_all[k], rew_all[k], done_all[k] = o, r, d
    obs_all, rew_all, done_all = obs_all.cpu().numpy(), rew_all.cpu().numpy(), done_all.cpu().numpy()
    err = np.zeros(steps)
    for k in range(steps):
        o_ref, r_ref, d_ref, _, _ = orc.step(kind, dt, T, case['task_c'], mem_ref, acts[k], case['dp'], flags=flags)
        run = length > k
        np.testing.assert_array_equal(o_ref[:, run], ref['obs'][k + 1][:, run])   # the replay IS the rollout
        eo = np.abs(obs_all[k][:, run] - o_ref[:, run])/np.maximum(1., np.abs(o_ref[:, run]))
        er = np.abs(rew_all[k][run] - r_ref[run])/np.maximum(1., np.abs(r_ref[run]))
        err[k] = max(eo.max(), er.max())
        # the episode ends on the device exactly where it ends in the oracle
        last = run & (length == k + 1)
        assert np.all(done_all[k][last] == 1) and np.all(done_all[k][run & ~last] == 0), f'done flags differ at step {k}'
        np.testing.assert_array_equal(done_all[k][run], d_ref[run])
    return err, length


def test_teacher_forced_benchmark_workload(eng, orc):
    """ QQube as bench.py runs it (dt = 1/500, T = 600, default randomizer's spread, FNN 6-64-64-1 with the reference's
    initialisation): 2048 environments, identical initial states, parameters and actions -> every step within 1e-9 """
    from bench import reference_fnn_params, qq_set_distrs
    kind, n, T = abi.ENV_QQ, 2048, 600
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    distrs, nominal = qq_set_distrs(0)
    dp = orc.randomize(n, distrs, nominal, seed=3)
    init = orc.sample_init(kind, n, dp, seed=3)
    ps = PolicySpec('fnn', O, A, hidden=[64, 64], hidden_nonlin=['tanh', 'tanh'])
    case = dict(kind=kind, dt=1/500., dp=dp, init=init, ps=ps, params=reference_fnn_params([O, 64, 64, A], 0).numpy(),
                task_c=envdefs.default_task_spec(kind).to_c(S, A))
    err, length = _teacher_forced(eng, orc, case, T)
    assert err.max() <= 1e-9, (float(err.max()), int(err.argmax()))
    assert length.max() >= 500 and length.min() < 300   # the mean episode of this workload lasts about 250 steps


@pytest.mark.parametrize('kind', cases.KINDS, ids=[cases.KIND_NAMES[k] for k in cases.KINDS])
def test_teacher_forced_every_kind(eng, orc, kind):
    """ up to 300 steps of every simulation under a small-gain policy: QQube, QCartPoleSwingUp, QBallBalancer,
    OneMassOscillator and Pendulum run the whole horizon, QCartPoleStab for 160 steps at the median; BallOnBeam and
    QQubeStab leave their state spaces after 70 and 40 steps at the median whatever a small policy does """
    T = 300
    case = cases.make_case(kind, 512, seed=400 + kind, policy='fnn', hidden=(16, 16), scale=0.05)
    err, length = _teacher_forced(eng, orc, case, T)
    assert length.max() >= 100
    assert err.max() <= 1e-9, (cases.KIND_NAMES[kind], float(err.max()), int(err.argmax()))


# ------------------------------------------------------------------------------------------------ the policy alone
def _torch_fnn(obs, params, sizes, nonlin, out_nonlin=None):
    """ FNN.forward (Pyrado/pyrado/policies/fnn.py:129-151) with torch.nn.functional.linear in fp64 """
    x = torch.as_tensor(obs, dtype=torch.float64)
    p = torch.as_tensor(params, dtype=torch.float64)
    nl = dict(tanh=torch.tanh, relu=torch.relu, sigmoid=torch.sigmoid)
    off = 0
    for j, (fi, fo) in enumerate(zip(sizes[:-1], sizes[1:])):
        W = p[off:off + fo*fi].reshape(fo, fi); off += fo*fi
        b = p[off:off + fo]; off += fo
        x = torch.nn.functional.linear(x, W, b)
        if j < len(sizes) - 2:
            x = nl[nonlin](x)
        elif out_nonlin is not None:
            x = nl[out_nonlin](x)
    return x.numpy()


@pytest.mark.parametrize('kind', [abi.ENV_QQ, abi.ENV_QCP_SU, abi.ENV_QBB, abi.ENV_PEND, abi.ENV_OMO],
                         ids=['qq-O6', 'qcp-O5', 'qbb-O8', 'pend-O3', 'omo-O2'])
@pytest.mark.parametrize('hidden', [(16,), (16, 16), (32, 32), (64, 64), (64, 64, 64, 64), (40, 24, 9)], ids=str)
def test_policy_forward_alone(eng, orc, kind, hidden):
    """ One-step rollouts from states spread over the whole state space: recorded action = policy(observation of the
    initial state).  (64, 64) runs with the fine tanh table, (64, 64, 64, 64) with the coarse one. """
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    rng = np.random.default_rng(77 + kind + len(hidden))
    n = 4096
    dp = np.tile(np.array(list(envdefs.nominal_domain_param(kind).values()))[:, None], (1, n))
    lo, hi = orc.constants(kind, dp[:, 0])[1:3]
    span = np.where(np.isfinite(hi), hi, 30.)
    init = rng.uniform(-1, 1, (S, n))*span[:, None]*0.999
    if kind in (abi.ENV_QQ_ST, abi.ENV_QCP_ST):
        init[1] += np.pi
    ps = PolicySpec('fnn', O, A, hidden=list(hidden), hidden_nonlin=['tanh']*len(hidden))
    params = rng.standard_normal(ps.num_param)*0.7
    task_c = envdefs.default_task_spec(kind).to_c(S, A)
    flags = abi.F_STOP_ON_DONE | (abi.F_SIMPLIFIED_DYN if kind == abi.ENV_QBB else 0)   # no IK needed for one action
    out = eng.rollout(kind, cases.DT[kind], 1, task_c, ps, params, init, dp, record=True, flags=flags)
    obs, act = out['obs'][0].cpu().numpy(), out['act'][0].cpu().numpy()          # [O, n], [A, n]
    assert np.isfinite(act).all()
    want_t = _torch_fnn(obs.T, params, [O] + list(hidden) + [A], 'tanh').T
    want_c = np.stack([orc.policy_forward(ps, params, obs[:, i]) for i in range(0, n, 8)], axis=1)
    e_t = np.max(np.abs(act - want_t)/np.maximum(1., np.abs(want_t)))
    e_c = np.max(np.abs(act[:, ::8] - want_c)/np.maximum(1., np.abs(want_c)))
    assert e_t <= 1e-12 and e_c <= 1e-12, (e_t, e_c)


@pytest.mark.parametrize('nonlin,out_nonlin', [('relu', None), ('sigmoid', 'tanh'), ('tanh', 'sigmoid')])
def test_policy_forward_alone_other_nonlinearities(eng, orc, nonlin, out_nonlin):
    kind = abi.ENV_QBB
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    rng = np.random.default_rng(5)
    n = 2048
    dp = np.tile(np.array(list(envdefs.nominal_domain_param(kind).values()))[:, None], (1, n))
    hi = orc.constants(kind, dp[:, 0])[2]
    init = rng.uniform(-1, 1, (S, n))*hi[:, None]*0.999
    ps = PolicySpec('fnn', O, A, hidden=[64, 32], hidden_nonlin=[nonlin]*2, out_nonlin=out_nonlin)
    params = rng.standard_normal(ps.num_param)*0.5
    out = eng.rollout(kind, 0.01, 1, envdefs.default_task_spec(kind).to_c(S, A), ps, params, init, dp, record=True,
                      flags=abi.F_STOP_ON_DONE | abi.F_SIMPLIFIED_DYN)
    obs, act = out['obs'][0].cpu().numpy(), out['act'][0].cpu().numpy()
    want = _torch_fnn(obs.T, params, [O, 64, 32, A], nonlin, out_nonlin).T
    assert np.max(np.abs(act - want)/np.maximum(1., np.abs(want))) <= 1e-12


@pytest.mark.parametrize('hidden', [(64, 64), (20, 7)], ids=str)
@pytest.mark.parametrize('nonlin', ['tanh', 'relu', 'sigmoid', None], ids=str)
def test_a_nan_weight_reaches_the_action_through_every_activation(eng, orc, nonlin, hidden):
    """ rollout() stops at a NaN action (sampling/rollout.py:190-201), so a NaN anywhere in the network has to come out of
    it: torch.relu keeps a NaN, and so must a relu written as a comparison (which by itself answers 0).  A NaN weight in
    each layer in turn; every rollout ends before its first step with the NaN status, as in the oracle. """
    kind, n, T = abi.ENV_QCP_SU, 300, 4
    case = cases.make_case(kind, n, seed=3, policy='fnn', hidden=hidden, nonlin=nonlin)
    ps = case['ps']
    sizes = [ps.obs_dim] + list(hidden) + [ps.act_dim]
    first = np.cumsum([0] + [sizes[l + 1]*sizes[l] + sizes[l + 1] for l in range(len(sizes) - 1)])[:-1]   # W_l[0][0]
    for l, off in enumerate(first):
        params = case['params'].copy()
        params[off + 1] = np.nan   # W_l[0][1]
        args = (kind, case['dt'], T, case['task_c'], ps, params, case['init'], case['dp'])
        ref = orc.rollout(*args, record=True)
        out = {k: v.cpu().numpy() for k, v in eng.rollout(*args, record=True).items()}
        assert np.all(ref['status'] == abi.ST_NAN) and np.all(ref['length'] == 0), l
        np.testing.assert_array_equal(out['status'], ref['status'], err_msg=str(l))
        np.testing.assert_array_equal(out['length'], ref['length'], err_msg=str(l))
        assert np.all(out['ret'] == 0.) and np.isnan(out['act']).all()


@pytest.mark.parametrize('kind', [abi.ENV_OMO, abi.ENV_QQ], ids=['omo', 'qq'])
@pytest.mark.parametrize('nonlin,hidden', [('tanh', (20, 7, 33)), ('sigmoid', (48, 64)), ('tanh', (64, 64))], ids=str)
def test_an_infinite_observation_saturates_a_network_of_padded_width(eng, orc, kind, nonlin, hidden):
    """ torch's tanh and sigmoid answer +-inf with +-1 and 1 / 0, so an infinite velocity in the observation still gives a
    finite action, and the rollout takes the step that turns its state into a NaN.  The rows that pad a layer to the MFMA
    tile must not make a NaN of that observation (a zero row would: 0 x inf): actions, rewards, lengths and status words as
    in the oracle, for environments with and without an infinite component side by side in one tile. """
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    n, T = 200, 3
    case = cases.make_case(kind, n, seed=12, policy='fnn', hidden=hidden, nonlin=nonlin)
    init = case['init'].copy()
    vel = S - 1                                   # the last state component is a velocity in both simulations
    init[vel, 5], init[vel, 70], init[vel, 71], init[vel, 199] = np.inf, -np.inf, np.inf, -np.inf
    for flags in (abi.F_STOP_ON_DONE, 0):
        args = (kind, case['dt'], T, case['task_c'], case['ps'], case['params'], init, case['dp'])
        ref = orc.rollout(*args, record=True, flags=flags)
        out = {k: v.cpu().numpy() for k, v in eng.rollout(*args, record=True, flags=flags).items()}
        assert np.isfinite(ref['act'][0, :, [5, 70, 71, 199]]).all() and np.all(ref['length'][[5, 70, 71, 199]] >= 1)
        np.testing.assert_array_equal(out['length'], ref['length'])
        np.testing.assert_array_equal(out['status'], ref['status'])
        for k in ('act', 'rew', 'ret'):
            assert cases.rel_err(out[k], ref[k]) <= 1e-9, k
