"""
The one trajectory on this path that the reference itself produced and ships:
Pyrado/data/perma/time_series/omo_traj_gt.npy, committed as tests/golden/ref_omo_traj_gt.npy (data only; provenance in
tests/golden/README.md).  Pyrado/scripts/sandbox/time_series/gen_ts_data.py:22-44,58,71 made it:
OneMassOscillatorSim(dt=0.01, max_steps=500) with m=1, k=10, d=2, init state [0, 0], a TimePolicy that pushes with 0.5
at t == 0 and with 0 afterwards, rollout(), and the position of all 501 observations.

CPU tests: the C oracle reproduces the 501 numbers bit for bit (reset + step with those actions, and its own
rollout loop), the NumPy one to the last bit but one (its BLAS fuses a multiply-add, see the test).  GPU tests: the same actions through psh_step, the same rollout through psh_rollout (an open-loop
action sequence as a LinearPolicy on a constant feature is not expressible, so the fused launch runs the equivalent
zero-gain policy from the state after the impulse), and through the host rollout() with the TimePolicy object, to
1e-9 relative per sample — the bar stated for floating-point parity.
"""
import functools
import os

import numpy as np
import pytest

from simurlacra_amd import _abi as abi, envdefs

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ref_omo_traj_gt.npy')
DT, T = 0.01, 500
DP = dict(m=1., k=10., d=2.)
RTOL = 1e-9


def golden():
    x = np.load(GOLDEN, allow_pickle=False)
    assert x.shape == (T + 1,) and x.dtype == np.float64
    return x


def actions():
    """ gen_ts_data.py:18-19 through TimePolicy.forward (policies/time.py:43-46): 0.5 on the first call only """
    a = np.zeros((T, 1, 1))
    a[0] = 0.5
    return a


def dp_column():
    names = list(envdefs.nominal_domain_param(abi.ENV_OMO).keys())
    assert names == ['m', 'k', 'd']
    return np.array([[DP[k]] for k in names])


def task_c():
    return envdefs.default_task_spec(abi.ENV_OMO).to_c(2, 1)


# ------------------------------------------------------------------------------------------------------ CPU
def test_fixture_is_the_force_excitation():
    x = golden()
    # after the impulse: xd = 0.5*dt/m, then x advances by xd*dt (forward Euler, one_mass_oscillator.py:75-84)
    assert x[0] == 0. and x[1] == 0. and x[2] == pytest.approx(0.5*DT*DT, rel=1e-15)
    assert np.abs(x).max() < 2e-3   # far inside the state space: the rollout runs its 500 steps


def test_c_oracle_reproduces_the_reference_trajectory_bit_for_bit():
    from oracle import oracle_lib as orc
    x = golden()
    dp = dp_column()
    mem = orc.reset(abi.ENV_OMO, DT, T, task_c(), np.zeros((2, 1)), dp)
    got = [mem['obs'][0, 0]]
    for k, a in enumerate(actions()):
        obs, rew, done, st, _ = orc.step(abi.ENV_OMO, DT, T, task_c(), mem, a, dp)
        got.append(obs[0, 0])
        last = k == T - 1      # the time limit ends the rollout; nothing fails on the way
        assert st[0] == (abi.ST_TIMEOUT if last else 0) and done[0] == int(last)
    np.testing.assert_array_equal(np.array(got), x)


def test_numpy_oracle_reproduces_the_reference_trajectory():
    """ The NumPy restatement makes the reference's own calls, `A.dot(state) + B.dot(act)` (one_mass_oscillator.py:79-81),
    and so inherits the BLAS of the NumPy it runs on: OpenBLAS 0.3.29's dgemv (NumPy 2.2 here) contracts
    `a10*x + a11*xd` into a fused multiply-add, the build that made the file rounded the product and the sum
    separately (the C oracle, compiled with -ffp-contract=off, does the same and is bit-equal).  Hence 1e-12, not
    equality, for this restatement; the second loop is the same recurrence in Python floats, which IS bit-equal. """
    from oracle import pysim_oracle_np as onp
    x = golden()
    env = onp.NpEnv('omo', DT, T)
    obs = env.reset(np.zeros(2), dict(DP))
    got = [obs[0]]
    for a in actions():
        obs, rew, done = env.step(a[:, 0].copy())[:3]
        got.append(obs[0])
    assert done
    np.testing.assert_allclose(np.array(got), x, rtol=1e-12, atol=0)

    omega, zeta = np.sqrt(DP['k']/DP['m']), DP['d']/(2.*np.sqrt(DP['m']*DP['k']))
    a10, a11 = -omega**2, -2.*zeta*omega
    s, plain = [0., 0.], [0.]
    for a in actions():
        sd = [0*s[0] + 1*s[1], (a10*s[0] + a11*s[1]) + (1./DP['m'])*a[0, 0]]
        s = [s[0] + sd[0]*DT, s[1] + sd[1]*DT]
        plain.append(s[0])
    np.testing.assert_array_equal(np.array(plain), x)


def test_c_oracle_rollout_loop_continues_the_reference_trajectory():
    """ The oracle's own rollout loop (the checker of every GPU rollout test) from the state after the impulse, under a
    LinearPolicy with zero gain = the TimePolicy's zero actions: samples 1..500 of the file, bit for bit. """
    from oracle import oracle_lib as orc
    from simurlacra_amd.specs import PolicySpec
    x = golden()
    s1 = np.array([[0.], [0.5*DT]])            # state after the first step: x unchanged, xd = dt*0.5/m
    ps = PolicySpec('linear', 2, 1, feats=['identity'])
    out = orc.rollout(abi.ENV_OMO, DT, T - 1, task_c(), ps, np.zeros(ps.num_param), s1, dp_column(), record=True)
    assert out['length'][0] == T - 1 and out['status'][0] == abi.ST_TIMEOUT
    np.testing.assert_array_equal(out['obs'][:, 0, 0], x[1:])


# ------------------------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope='module')
def eng():
    from simurlacra_amd.engine import get_engine
    return get_engine(0)


def _assert_close(got, want):
    got, want = np.asarray(got), np.asarray(want)
    err = np.abs(got - want)/np.maximum(np.abs(want), 1e-300)
    # relative to each sample's own magnitude (|x| <= 1.1e-3 here), not to 1: 1e-9 of the sample
    nz = want != 0.
    assert np.all(got[~nz] == 0.)
    assert err[nz].max() <= RTOL, (float(err[nz].max()), int(err[nz].argmax()))
    return float(err[nz].max())


@pytest.mark.gpu
def test_device_step_reproduces_the_reference_trajectory(eng):
    """ psh_reset + 500 psh_step launches with the TimePolicy's actions, a batch of 64 identical environments (every
    lane of a wave must give the same numbers) """
    import torch
    x = golden()
    n = 64
    dp = eng.f64(np.tile(dp_column(), (1, n)))
    mem = eng.reset(abi.ENV_OMO, DT, T, task_c(), np.zeros((2, n)), dp)
    rows = [mem['obs'][0].clone()]
    acts = eng.f64(np.tile(actions(), (1, 1, n)))
    done_last = None
    for k in range(T):
        obs, rew, done, st = eng.step(abi.ENV_OMO, DT, T, task_c(), mem, acts[k].contiguous(), dp)
        rows.append(obs[0].clone())
        done_last = done
    got = torch.stack(rows).cpu().numpy()          # [501, n]
    assert np.all(got == got[:, :1])
    assert np.all(done_last.cpu().numpy() == 1)
    worst = _assert_close(got[:, 0], x)
    assert worst <= 1e-12                          # observed: a few ulp (FMA contraction against NumPy's separate ops)


@pytest.mark.gpu
def test_device_rollout_continues_the_reference_trajectory(eng):
    """ the fused rollout kernel (reset, in-kernel loop, trajectory writeback) from the state after the impulse under a
    zero-gain LinearPolicy and a zero FNNPolicy: samples 1..500 of the file """
    from simurlacra_amd.specs import PolicySpec
    x = golden()
    n = 200
    s1 = np.tile(np.array([[0.], [0.5*DT]]), (1, n))
    dp = np.tile(dp_column(), (1, n))
    for ps in (PolicySpec('linear', 2, 1, feats=['identity']),
               PolicySpec('fnn', 2, 1, hidden=[16, 16], hidden_nonlin=['tanh', 'tanh'])):
        out = eng.rollout(abi.ENV_OMO, DT, T - 1, task_c(), ps, np.zeros(ps.num_param), s1, dp, record=True)
        assert np.all(out['length'].cpu().numpy() == T - 1) and np.all(out['status'].cpu().numpy() == abi.ST_TIMEOUT)
        obs = out['obs'].cpu().numpy()[:, 0, :]    # [500, n]
        assert np.all(obs == obs[:, :1])
        _assert_close(obs[:, 0], x[1:])


@pytest.mark.gpu
def test_host_rollout_with_time_policy_reproduces_the_reference_trajectory(eng):
    """ gen_ts_data.py:31-44 written against this package: the same calls, the same 501 numbers """
    from simurlacra_amd.environments.pysim.one_mass_oscillator import OneMassOscillatorSim
    from simurlacra_amd.policies.time import TimePolicy
    from simurlacra_amd.sampling.rollout import rollout
    x = golden()

    def dirac_impulse(t, env_spec, amp):
        return amp*np.ones(env_spec.act_space.shape) if t == 0 else np.zeros(env_spec.act_space.shape)

    env = OneMassOscillatorSim(DT, np.ceil(5/DT))
    env.domain_param = dict(m=1., k=10., d=2.0)
    policy = TimePolicy(env.spec, functools.partial(dirac_impulse, env_spec=env.spec, amp=0.5), DT)
    ro = rollout(env, policy, reset_kwargs=dict(init_state=np.array([0, 0])), record_dts=False)
    assert ro.length == T and ro.observations.shape == (T + 1, 2)
    _assert_close(ro.observations[:, 0], x)
    assert ro.actions[0, 0] == 0.5 and np.all(ro.actions[1:] == 0.)
