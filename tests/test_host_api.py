"""
CPU tests of the host-side mirror of the Pyrado interface: spaces, tasks, domain randomization, wrappers, policies
and StepSequence.  They follow the reference's own tests for these parts (cited per test); nothing here computes a
simulation step (that needs the GPU).
"""
import numpy as np
import pytest
import torch as to

import simurlacra_amd as pyrado
from simurlacra_amd import _abi as abi
from simurlacra_amd.domain_randomization import (BernoulliDomainParam, DomainRandomizer, MultivariateNormalDomainParam,
                                                 NormalDomainParam, UniformDomainParam, get_default_randomizer,
                                                 get_default_randomizer_qbb, get_zero_var_randomizer)
from simurlacra_amd.environment_wrappers import (GaussianActNoiseWrapper, GaussianObsNoiseWrapper, ObsNormWrapper,
                                                 ObsPartialWrapper)
from simurlacra_amd.environment_wrappers import (ActDelayWrapper, ActNormWrapper, DomainRandWrapperBuffer,
                                                 DomainRandWrapperLive, MetaDomainRandWrapper, inner_env,
                                                 remove_all_dr_wrappers, typed_env)
from simurlacra_amd.environments import (BallOnBeamSim, OneMassOscillatorSim, PendulumSim, QBallBalancerSim,
                                         QCartPoleStabSim, QCartPoleSwingUpSim, QQubeSim, QQubeStabSim, QQubeSwingUpSim,
                                         SimEnv)
from simurlacra_amd.policies import (FeatureStack, FNNPolicy, LinearPolicy, const_feat, identity_feat, squared_feat)
from simurlacra_amd.sampling import ParallelRolloutSampler, ParallelSampler, StepSequence
from simurlacra_amd.spaces import BoxSpace, CompoundSpace, Polar2DPosVelSpace
from simurlacra_amd.tasks import (DesStateTask, FinalRewMode, FinalRewTask, QuadrErrRewFcn, RadiallySymmDesStateTask,
                                  ScaledExpQuadrErrRewFcn, task_to_spec)
from simurlacra_amd.utils.data_types import EnvSpec

ENVS = [lambda: QQubeSim(dt=0.004, max_steps=100), lambda: QCartPoleSwingUpSim(dt=0.002, max_steps=100),
        lambda: QCartPoleStabSim(dt=0.002, max_steps=100), lambda: QBallBalancerSim(dt=0.01, max_steps=100),
        lambda: BallOnBeamSim(dt=0.01, max_steps=100), lambda: OneMassOscillatorSim(dt=0.01, max_steps=100),
        lambda: PendulumSim(dt=0.02, max_steps=100), lambda: QQubeStabSim(dt=0.004, max_steps=100)]


class MockEnv(SimEnv):
    """ Records the last action it was stepped with (as Pyrado/tests/environment_wrappers/mock_env.py does). """

    def __init__(self, obs_space=None, act_space=None):
        self._obs_space, self._act_space, self._domain_param = obs_space, act_space, {}
        self.last_act = None

    obs_space = property(lambda self: self._obs_space)
    state_space = property(lambda self: self._obs_space)
    act_space = property(lambda self: self._act_space)
    init_space = property(lambda self: None)
    task = property(lambda self: None)

    @property
    def domain_param(self):
        return self._domain_param.copy()

    @domain_param.setter
    def domain_param(self, param):
        self._domain_param = dict(param)

    @classmethod
    def get_nominal_domain_param(cls):
        return {}

    def reset(self, init_state=None, domain_param=None):
        if domain_param is not None:
            self.domain_param = domain_param
        return None

    def step(self, act):
        self.last_act = list(act)
        return None, 0., False, dict()


# ------------------------------------------------------------------------------------------------- spaces
def test_torus2D():
    """ Pyrado/tests/test_spaces.py:145-154 """
    s0 = Polar2DPosVelSpace(np.array([1, 0, -0.1, -0.1]), np.array([1, 0, 0.1, 0.1])).sample_uniform()
    assert s0[0] == 1 and s0[1] == 0
    s90 = Polar2DPosVelSpace(np.array([1, np.pi/2, 0, 0]), np.array([1, np.pi/2, 0, 0])).sample_uniform()
    assert np.all(np.isclose(s90, np.array([0, 1, 0, 0])))


def test_box_space_contains_project_sample():
    bs = BoxSpace([-1, -2], [1, np.inf], labels=['a', 'b'])
    assert bs.contains(np.array([1., -2.])) and not bs.contains(np.array([1.0001, 0.]))
    inside = np.array([0.5, 3.])
    assert bs.project_to(inside) is inside                       # same object when already inside
    np.testing.assert_array_equal(bs.project_to(np.array([2., -5.])), [1., -2.])
    with pytest.raises(pyrado.ValueErr):
        bs.contains(np.array([np.nan, 0.]))
    with pytest.raises(pyrado.ShapeErr):
        bs.contains(np.zeros(3))
    for _ in range(20):
        assert bs.contains(bs.sample_uniform())
    np.testing.assert_array_equal(BoxSpace(-1, 1, shape=(3,)).bound_abs_up, np.ones(3))
    cs = CompoundSpace([BoxSpace([-2.], [-1.]), BoxSpace([1.], [2.])])
    assert all(1 <= abs(cs.sample_uniform()[0]) <= 2 for _ in range(20))


# -------------------------------------------------------------------------------------------------- tasks
def test_modulated_rew_fcn():
    """ Pyrado/tests/test_tasks.py:44-61 """
    Q, R = np.eye(4), np.eye(2)
    s, a = np.array([1., 2, 3, 4]), np.array([0., 0])
    rew_fcn = QuadrErrRewFcn(Q, R)
    task = RadiallySymmDesStateTask(EnvSpec(None, None, None), np.zeros(4), rew_fcn, [0, 1, 3], 2)
    assert task.step_rew(s, a) == -(1**2 + 3**2)
    task = RadiallySymmDesStateTask(EnvSpec(None, None, None), np.zeros(4), rew_fcn, [1, 3], np.array([2, 3]))
    assert task.step_rew(s, a) == -(1**2 + 3**2 + 1**2)


def test_final_rew_task_yields_once_and_modes():
    """ FinalRewTask gives its reward once (Pyrado/tests/test_tasks.py:181-188,252-259); failure branches of
    final_reward.py:87-183 """
    spec = EnvSpec(obs_space=BoxSpace(-1, 1, 3), act_space=BoxSpace(-1, 1, 2), state_space=BoxSpace(-1, 1, 3))
    inner = DesStateTask(spec, np.zeros(3), QuadrErrRewFcn(np.eye(3), np.eye(2)))
    out = 2*np.ones(3)
    t = FinalRewTask(inner, FinalRewMode(always_negative=True), factor=100)
    assert t.final_rew(out, 0) == pytest.approx(-100.) and t.final_rew(out, 0) == pytest.approx(0.)
    t.reset(env_spec=spec)
    assert t.final_rew(np.zeros(3), 5) == 0.            # neither failed nor succeeded
    t = FinalRewTask(inner, FinalRewMode(state_dependent=True, time_dependent=True))
    assert t.final_rew(out, 7) == pytest.approx(-7*12.)
    t = FinalRewTask(inner, FinalRewMode(always_positive=True))
    assert t.final_rew(out, 7) == 0.
    t = FinalRewTask(inner, FinalRewMode(always_positive=True, always_negative=True))
    with pytest.raises(NotImplementedError):
        t.final_rew(out, 7)


def test_scaled_exp_reward_rescales_with_the_spaces():
    ss, as_ = BoxSpace(-np.ones(2), np.ones(2)), BoxSpace(-np.ones(1), np.ones(1))
    rf = ScaledExpQuadrErrRewFcn(np.eye(2), np.eye(1), ss, as_, min_rew=1e-4)
    assert rf(np.ones(2), np.ones(1)) == pytest.approx(1e-4)
    rf.reset(BoxSpace(-2*np.ones(2), 2*np.ones(2)), as_)
    assert rf(2*np.ones(2), np.ones(1)) == pytest.approx(1e-4)


@pytest.mark.parametrize('make', ENVS)
def test_default_tasks_translate_to_the_kernel_spec(make):
    env = make()
    spec = env.task_spec()
    S, O, A, P, _ = abi.ENV_DIMS[env.kind]
    c = spec.to_c(S, A)
    assert env.obs_space.flat_dim == O and env.act_space.flat_dim == A and env.state_space.flat_dim == S
    assert list(env.get_nominal_domain_param().keys()) == list(env.supported_domain_param)
    assert len(env.domain_param_column()) == P
    if env.kind == abi.ENV_QQ:
        assert c.rew_kind == abi.REW_EXP_QUADR and c.radial == 1 and c.radial_mask == 2 and c.final_mode == 0
    if env.kind == abi.ENV_QCP_SU:
        assert c.rew_kind == abi.REW_SWINGUP and c.final_mode == abi.FIN_ENABLED | abi.FIN_ALWAYS_NEGATIVE
    if env.kind == abi.ENV_QCP_ST:
        assert c.final_mode == abi.FIN_ENABLED | abi.FIN_STATE_DEPENDENT | abi.FIN_TIME_DEPENDENT
    if env.kind == abi.ENV_OMO:
        assert c.rew_kind == abi.REW_QUADR and c.final_factor == 1e3
    if env.kind == abi.ENV_PEND:      # the radial index of the reference's pendulum task is the velocity
        assert c.rew_kind == abi.REW_EXP_QUADR and c.radial_mask == 2 and c.state_des[0] == np.pi
        assert np.all(env.init_space.sample_uniform() == 0.) and env.act_space.bound_up[0] == 3.5
    if env.kind == abi.ENV_QQ_ST:
        assert c.Q[0] == 3. and c.R[0] == 5e-2 and env.obs_space is env.state_space
        np.testing.assert_allclose(env.observe(np.array([0.1, 3., 1., 2.])), [-0.1, np.pi - 3., -1., -2.])


def test_unsupported_tasks_are_rejected():
    spec = EnvSpec(BoxSpace(-1, 1, 3), BoxSpace(-1, 1, 2), BoxSpace(-1, 1, 3))
    inner = DesStateTask(spec, np.zeros(3), QuadrErrRewFcn(np.eye(3), np.eye(2)))
    with pytest.raises(NotImplementedError):
        task_to_spec(FinalRewTask(FinalRewTask(inner, FinalRewMode()), FinalRewMode()))
    with pytest.raises(NotImplementedError):
        task_to_spec(DesStateTask(spec, np.zeros(3), QuadrErrRewFcn(np.eye(3), np.eye(2)), success_fcn=lambda e: True))


def test_names_of_later_upstream_versions_are_exported():
    assert QQubeSwingUpSim is QQubeSim and ParallelRolloutSampler is ParallelSampler


def test_spaces_follow_the_domain_parameters():
    """ action/state bounds that depend on randomized parameters (pysim/ball_on_beam.py:22-27 etc.) """
    env = BallOnBeamSim(dt=0.01, max_steps=10)
    env.domain_param = dict(l_beam=3., g=10.)
    assert env.act_space.bound_up[0] == pytest.approx(3./2*10*3) and env.state_space.bound_up[0] == 1.5
    c1 = env.task.rew_fcn.c_max
    env.domain_param = dict(l_beam=1.)
    assert env.task.rew_fcn.c_max != c1
    omo = OneMassOscillatorSim(dt=0.01, max_steps=10)
    omo.domain_param = dict(k=12.)
    assert omo.act_space.bound_up[0] == 12.
    qcp = QCartPoleSwingUpSim(dt=0.01, max_steps=10, long=True)
    assert qcp.domain_param['m_pole'] == 0.23 and qcp.state_space.bound_up[0] == pytest.approx(0.814/2 - 0.05)


# ---------------------------------------------------------------------------------------- domain randomization
@pytest.mark.parametrize('dp', [
    NormalDomainParam(name='', mean=10, std=1., clip_lo=9, clip_up=11),
    MultivariateNormalDomainParam(name='', mean=to.ones((2, 1), dtype=to.float64), cov=to.eye(2, dtype=to.float64), clip_lo=-1, clip_up=1.),
    BernoulliDomainParam(name='', val_0=2, val_1=5, prob_1=0.8),
    BernoulliDomainParam(name='', val_0=-3, val_1=5, prob_1=0.8, clip_up=4),
    UniformDomainParam(name='', mean=0., halfspan=0.),
], ids=['1dim', '2dim', 'bern_v1', 'bern_v2', 'zero_span'])
def test_domain_param(dp):
    """ Pyrado/tests/test_domain_randomization.py:13-26 """
    for num_samples in [1, 5, 25]:
        s = dp.sample(num_samples)
        assert len(s) == num_samples
        assert all(float(v.max()) <= dp.clip_up and float(v.min()) >= dp.clip_lo for v in s)


def test_randomizer_formats():
    """ Pyrado/tests/test_domain_randomization.py:38-93 """
    r = DomainRandomizer(NormalDomainParam(name='mass', mean=1.2, std=0.1, clip_lo=10, clip_up=100),
                         UniformDomainParam(name='special', mean=0, halfspan=42, clip_lo=-7.4, roundint=True),
                         NormalDomainParam(name='length', mean=4, std=0.6, clip_up=50.1),
                         MultivariateNormalDomainParam(name='multidim', mean=10*to.ones((2,), dtype=to.float64),
                                                       cov=2*to.eye(2, dtype=to.float64), clip_up=11))
    r.randomize(7)
    for n, count in ((3, 3), (-1, 7)):
        for dtype, tp in (('torch', to.Tensor), ('numpy', np.ndarray)):
            d = r.get_params(n, format='dict', dtype=dtype)
            assert isinstance(d, dict) and isinstance(d['mass'], list) and len(d['mass']) == count
            assert isinstance(d['mass'][0], tp) and np.asarray(d['multidim'][0]).size == 2
            lst = r.get_params(n, format='list', dtype=dtype)
            assert isinstance(lst, list) and len(lst) == count and isinstance(lst[0], dict)
            assert isinstance(lst[0]['mass'], tp) and np.asarray(lst[0]['multidim']).size == 2
    assert float(r.get_params(1)['mass']) >= 10
    special = r.get_params(-1, format='dict', dtype='torch')['special']
    assert all(v.dtype == to.int for v in special)


def test_rescaling_and_zero_variance():
    """ Pyrado/tests/test_domain_randomization.py:96-105 """
    env = QQubeSim(dt=0.01, max_steps=10)
    r0, r = get_default_randomizer(env), get_default_randomizer(env)
    r.rescale_distr_param('std', 12.5)
    assert r.domain_params[0].std == 12.5*r0.domain_params[0].std and r.domain_params[0].mean == r0.domain_params[0].mean
    z = get_zero_var_randomizer(env)
    z.randomize(5)
    for d in z.get_params(-1, 'list', 'numpy'):
        assert d['Mp'] == pytest.approx(0.024)


def test_default_qbb_randomizer_keeps_the_reference_quirks():
    """ default_randomizers.py:192-213: gravity entry is named ']' (never applied), negative half spans on the
    negative voltage thresholds sample the same interval """
    r = get_default_randomizer_qbb()
    assert r.domain_params[0].name == ']'
    env = QBallBalancerSim(dt=0.01, max_steps=10)
    rows = [d['row'] for d in r.device_table(list(env.get_nominal_domain_param().keys()))]
    assert 0 not in rows and len(rows) == 19
    r.randomize(200)
    v = np.array([float(d['V_thold_x_neg']) for d in r.get_params(-1, 'list', 'numpy')])
    assert np.all((v >= -0.1 - 0.1/3) & (v <= -0.1 + 0.1/3))


def test_dr_wrappers_and_chain_utils():
    """ ring buffer semantics of environment_wrappers/domain_randomization.py:151-199; wrapper removal :212-229 """
    env = BallOnBeamSim(dt=0.01, max_steps=10)
    rand = get_default_randomizer(env)
    buf = DomainRandWrapperBuffer(env, rand)
    buf.fill_buffer(4)
    assert len(buf.buffer) == 4 and buf.ring_idx == 0
    buf.ring_idx = 3
    assert buf.ring_idx == 3
    meta = MetaDomainRandWrapper(DomainRandWrapperLive(env, rand), {0: ('m_ball', 'mean'), 1: ('m_ball', 'std')})
    meta.adapt_randomizer(np.array([0.7, 0.01]))
    mb = [p for p in meta.randomizer.domain_params if p.name == 'm_ball'][0]
    assert mb.mean == 0.7 and mb.std == 0.01 and float(mb.distr.mean) == pytest.approx(0.7)
    assert inner_env(meta) is env and typed_env(meta, DomainRandWrapperLive) is meta.wrapped_env
    assert remove_all_dr_wrappers(ActNormWrapper(meta)).wrapped_env is env


# ------------------------------------------------------------------------------------------------ wrappers
def test_act_delay():
    """ Pyrado/tests/environment_wrappers/test_action_delay.py:24-89 """
    mockenv = MockEnv(act_space=BoxSpace(-1, 1, shape=(2,)))
    wenv = ActDelayWrapper(mockenv, delay=2)
    wenv.reset()
    for act, exp in (([0, 1], [0, 0]), ([2, 4], [0, 0]), ([1, 2], [0, 1]), ([2, 3], [2, 4])):
        wenv.step(np.array(act))
        assert mockenv.last_act == exp
    wenv = ActDelayWrapper(mockenv, delay=1)
    wenv.reset()
    wenv.step(np.array([0, 4])); assert mockenv.last_act == [0, 0]
    wenv.step(np.array([4, 4])); assert mockenv.last_act == [0, 4]
    wenv.reset()
    wenv.step(np.array([1, 2])); assert mockenv.last_act == [0, 0]
    wenv.domain_param = {'act_delay': 2}
    wenv.reset()
    for act, exp in (([1, 2], [0, 0]), ([2, 3], [0, 0]), ([8, 9], [1, 2])):
        wenv.step(np.array(act))
        assert mockenv.last_act == exp
    assert wenv.domain_param['act_delay'] == 2
    nodelay = ActDelayWrapper(mockenv, delay=0)
    nodelay.reset()
    nodelay.step(np.array([4, 1])); assert mockenv.last_act == [4, 1]


def test_act_norm():
    """ Pyrado/tests/environment_wrappers/test_action_normalization.py:9-37 """
    mockenv = MockEnv(act_space=BoxSpace([-2, -1, 0], [2, 3, 1]))
    uut = ActNormWrapper(mockenv)
    lb, ub = uut.act_space.bounds
    assert np.all(lb == -1) and np.all(ub == 1)
    for act, exp in (([0, 0, 0], [0, 1, 0.5]), ([1, 1, 1], [2, 3, 1]), ([-1, -1, -1], [-2, -1, 0])):
        uut.step(np.array(act))
        assert mockenv.last_act == exp


def test_obs_wrappers_and_noise_wrappers():
    """ ObsNormWrapper / ObsPartialWrapper / Gaussian noise wrappers as Pyrado/pyrado/environment_wrappers/
    {observation_normalization.py:73-106, observation_partial.py:10-47, observation_noise.py, action_noise.py} define them """
    space = BoxSpace([-2., -1., -np.inf], [2., 3., np.inf], labels=['a', 'b', 'c'])
    mockenv = MockEnv(obs_space=space, act_space=BoxSpace(-1, 1, shape=(2,)))
    with pytest.raises(pyrado.ValueErr):
        ObsNormWrapper(mockenv).obs_space                                   # infinite bound, not overridden
    with pytest.raises(pyrado.ValueErr):
        ObsNormWrapper(mockenv, explicit_lb={'a': -4.})._process_obs(np.zeros(3))   # override dict given, 'c' still infinite
    norm = ObsNormWrapper(mockenv, explicit_lb={'c': -10.}, explicit_ub={'c': 10., 'a': 4.})
    np.testing.assert_allclose(norm._process_obs(np.array([1., 1., 5.])), [0., 0., 0.5])
    lo, hi = norm.obs_space.bounds
    # the reported space keeps the original (here infinite) extent of an overridden entry, in normalised units
    np.testing.assert_allclose(lo, [-1, -1, -np.inf]); np.testing.assert_allclose(hi, [1/3, 1, np.inf])
    part = ObsPartialWrapper(mockenv, idcs=['b'])
    assert list(part.obs_space.labels) == ['a', 'c'] and list(part._process_obs(np.array([1., 2., 3.]))) == [1., 3.]
    part = ObsPartialWrapper(mockenv, mask=[0, 1, 1], keep_selected=True)
    assert list(part._process_obs(np.array([1., 2., 3.]))) == [2., 3.] and part.obs_space.flat_dim == 2
    with pytest.raises(pyrado.ShapeErr):
        ObsPartialWrapper(mockenv, mask=[0, 1])
    on = GaussianObsNoiseWrapper(mockenv, noise_std=[0., 0., 0.], noise_mean=[1., 2., 3.])
    np.testing.assert_allclose(on._process_obs(np.zeros(3)), [1., 2., 3.])
    on.domain_param = dict(obs_noise_mean=[0., 0., 1.])
    np.testing.assert_allclose(on._process_obs(np.zeros(3)), [0., 0., 1.])
    assert 'obs_noise_std' in on.domain_param
    an = GaussianActNoiseWrapper(mockenv, noise_mean=[0.5, -0.5])
    an.step(np.array([1., 1.])); assert mockenv.last_act == [1.5, 0.5]
    an.domain_param = dict(act_noise_mean=[9., 9.])      # the reference stores these where _process_act never looks
    an.step(np.array([1., 1.])); assert mockenv.last_act == [1.5, 0.5]
    with pytest.raises(pyrado.ShapeErr):
        GaussianActNoiseWrapper(mockenv, noise_std=[1.])


def test_env_plan_translates_the_wrapper_chain():
    """ the chain of scripts/training/qbb_spota_hc.py:29-38 as the kernels take it """
    from simurlacra_amd.domain_randomization import UniformDomainParam, get_default_randomizer
    from simurlacra_amd.environment_wrappers import DomainRandWrapperLive
    from simurlacra_amd.sampling.batch import EnvPlan
    env = QBallBalancerSim(dt=0.01, max_steps=50)
    env = GaussianObsNoiseWrapper(env, noise_std=[1/180*np.pi, 1/180*np.pi, 0.005, 0.005, 10/180*np.pi, 10/180*np.pi, 0.05, 0.05])
    env = ActNormWrapper(env)
    env = ActDelayWrapper(env)
    randomizer = get_default_randomizer(env)
    randomizer.add_domain_params(UniformDomainParam(name='act_delay', mean=5, halfspan=5, clip_lo=0, roundint=True))
    env = DomainRandWrapperLive(env, randomizer)
    plan = EnvPlan(env)
    spec = plan.wrap_spec(noise_seed=3, env_offset=7)
    assert list(spec.ops) == ['act_delay', 'act_norm', 'obs_noise'] and spec.act_delay_row == 20 and plan.obs_dim == 8
    c = spec.to_c(8, 2)
    assert c.n_ops == 3 and list(c.ops)[:3] == [abi.WRAP_ACT_DELAY, abi.WRAP_ACT_NORM, abi.WRAP_OBS_NOISE]
    assert c.obs_noise_std[7] == 0.05 and c.env_offset == 7 and np.isnan(c.obs_lo[0])
    assert [d['row'] for d in randomizer.device_table(plan.names, plan.extra_rows)][-1] == 20
    # partial observation outside the noise: vectors are re-indexed by the simulation's observation index
    # QQube's velocities are unbounded: without overrides for both the reference refuses to normalise, and so does the plan
    with pytest.raises(pyrado.ValueErr):
        EnvPlan(ObsNormWrapper(QQubeSim(dt=0.004, max_steps=10), explicit_lb={r'$\dot{\theta}$': -20.})).wrap_spec()
    env = ObsPartialWrapper(GaussianObsNoiseWrapper(QQubeSim(dt=0.004, max_steps=10), noise_std=[.1, .2, .3, .4, .5, .6]), idcs=[1])
    spec = EnvPlan(env).wrap_spec()
    assert list(spec.ops) == ['obs_partial', 'obs_noise'] and spec.obs_dim(6) == 5 and list(spec.obs_keep) == [1, 0, 1, 1, 1, 1]
    with pytest.raises(NotImplementedError):
        EnvPlan(ActNormWrapper(ActNormWrapper(QQubeSim(dt=0.004, max_steps=10))))


# ------------------------------------------------------------------------------------------------ policies
def test_policies_param_round_trip_and_fused_spec():
    """ param_values round trip and batched-vs-single forward (Pyrado/tests/test_policies.py:316,460-461) """
    env = QQubeSim(dt=0.004, max_steps=10)
    lin = LinearPolicy(env.spec, FeatureStack([const_feat, identity_feat, squared_feat]))
    fnn = FNNPolicy(env.spec, [64, 64], to.tanh)
    for pol, n in ((lin, 13), (fnn, 6*64 + 64 + 64*64 + 64 + 64 + 1)):
        assert pol.num_param == n == pol.fused_spec().num_param
        p = to.randn(n, dtype=to.float64)
        pol.param_values = p
        assert to.equal(pol.param_values, p)
        obs = to.randn(5, 6, dtype=to.float64)
        batched = pol(obs)
        assert batched.shape == (5, 1)
        for i in range(5):
            assert to.allclose(pol(obs[i]), batched[i], rtol=1e-12, atol=1e-14)
    w = fnn.net.hidden_layers[1].weight.data
    fnn.init_param()
    w = fnn.net.hidden_layers[1].weight.data
    assert to.allclose(w @ w.t(), to.eye(64, dtype=to.float64), atol=1e-10)     # orthogonal initialisation
    assert LinearPolicy(env.spec, FeatureStack([lambda x: x])).fused_spec() is None
    assert FNNPolicy(env.spec, [8], to.nn.functional.elu).fused_spec() is None


def test_oracle_agrees_with_the_policy_modules():
    """ the oracle's policy forward equals the torch modules the host layer hands to users """
    from oracle import oracle_lib as orc
    env = QBallBalancerSim(dt=0.01, max_steps=10)
    obs = np.random.default_rng(0).standard_normal(8)
    for pol in (LinearPolicy(env.spec, FeatureStack([const_feat, identity_feat, squared_feat])),
                FNNPolicy(env.spec, [32, 16], to.tanh, output_nonlin=to.tanh)):
        with to.no_grad():
            ref = pol(to.as_tensor(obs)).numpy()
        got = orc.policy_forward(pol.fused_spec(), pol.param_values.detach().numpy(), obs)
        np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-13)


# -------------------------------------------------------------------------------------------- StepSequence
def test_step_sequence_basics():
    """ construction, truncation, slicing, concat, returns (Pyrado/tests/test_rollout_result.py) """
    obs = [np.array([i, i + 0.5]) for i in range(6)]
    act = [np.array([float(i)]) for i in range(5)]
    rew = [1., 2., 3., 4., 5.]
    infos = [dict(t=0.1*i) for i in range(5)]
    ro = StepSequence(observations=obs, actions=act, rewards=rew, env_infos=infos, rollout_info=dict(x=1))
    assert len(ro) == 5 and ro.observations.shape == (6, 2) and ro.actions.shape == (5, 1) and ro.done[-1]
    assert ro.get_data_values('observations', truncate_last=True).shape == (5, 2)
    assert ro.undiscounted_return() == 15. and ro.discounted_return(0.5) == pytest.approx(1 + 1 + .75 + .5 + 5/16)
    assert ro[2].reward == 3. and ro[2].next_observation[0] == 3 and ro[-1].done
    assert ro.env_infos['t'].shape == (5,)
    part = ro[1:4]
    assert len(part) == 3 and part.observations.shape == (4, 2) and part.rewards.tolist() == [2., 3., 4.]
    both = StepSequence.concat([ro, ro])
    assert len(both) == 10 and both.rollout_count == 2 and list(both.rollout_lengths) == [5, 5]
    assert both.get_rollout(1).rewards.tolist() == rew
    with pytest.raises(pyrado.ShapeErr):
        both.undiscounted_return()
    ro.torch()
    assert isinstance(ro.rewards, to.Tensor) and float(ro.undiscounted_return()) == 15.
    ro.numpy()
    assert isinstance(ro.observations, np.ndarray)
    with pytest.raises(ValueError):
        StepSequence(observations=[], actions=[], rewards=[])
    with pytest.raises(ValueError):
        StepSequence(rewards=rew, actions=act)


def test_structured_features():
    """ MultFeat / RandFourierFeat / RBFFeat (Pyrado/pyrado/policies/features.py:143-303): sizes, values, and a
    LinearPolicy on top of them, which the kernels take as FeatSpecs with the features' data """
    from simurlacra_amd.policies import MultFeat, RandFourierFeat, RBFFeat
    x = to.tensor([0.5, -2., 3.], dtype=to.float64)
    xb = to.stack([x, 2*x])
    mf = MultFeat([0, 2])
    assert mf(x).tolist() == [1.5] and mf(xb).shape == (2, 1) and mf(xb)[1, 0] == 6.
    with pytest.raises(pyrado.ShapeErr):
        MultFeat([1])
    to.manual_seed(0)
    rff = RandFourierFeat(inp_dim=3, num_feat_per_dim=7, bandwidth=np.array([1., 2., 4.]))
    ref = np.sqrt(2/7.)*np.cos(rff.freq.numpy() @ x.numpy() + rff.shift.numpy())
    np.testing.assert_allclose(rff(x).numpy(), ref, rtol=1e-12)
    assert rff(xb).shape == (2, 7)
    rbf = RBFFeat(num_feat_per_dim=5, bounds=(np.array([-1., -3., 0.]), np.array([1., 3., 4.])))
    v = rbf(x)
    assert v.shape == (15,) and np.allclose(v.reshape(3, 5).sum(dim=1).numpy(), 1.)
    c0 = np.linspace(-1, 1, 5)
    e0 = np.exp(np.log(0.2)/0.25*(0.5 - c0)**2)
    np.testing.assert_allclose(v[:5].numpy(), e0/e0.sum(), rtol=1e-12)
    joint = RBFFeat(num_feat_per_dim=5, bounds=(np.array([-1., -3., 0.]), np.array([1., 3., 4.])), state_wise_norm=False)
    assert joint(xb).shape == (2, 15) and np.allclose(joint(xb).sum(dim=1).numpy(), 1.)
    with pytest.raises(pyrado.ValueErr):
        RBFFeat(num_feat_per_dim=1, bounds=(np.zeros(2), np.ones(2)))
    stack = FeatureStack([const_feat, identity_feat, mf, rff, rbf])
    assert stack.get_num_feat(3) == 1 + 3 + 1 + 7 + 15 and stack(x).shape == (27,) and stack(xb).shape == (2, 27)
    names = stack.fused_names()
    assert names[:2] == ['const', 'identity'] and [f.kind for f in names[2:]] == ['mult', 'rff', 'rbf']
    env = QQubeSim(dt=0.004, max_steps=10)
    pol = LinearPolicy(env.spec, FeatureStack([identity_feat, RBFFeat(3, env.obs_space.bounds)]))
    spec = pol.fused_spec()
    assert pol.num_param == 6 + 18 and spec.num_param == 24 and spec.feat_data().size == 6 + 18
    pc = spec.to_c(feat_data_ptr=1)
    assert list(pc.feats)[:2] == [1, 14] and pc.feat_size[1] == 3 and pc.feat_arg[1] == 1 and pc.feat_data_len == 24
    # what the kernels do not take stays with the torch step loop
    assert LinearPolicy(env.spec, FeatureStack([RBFFeat(100, env.obs_space.bounds)])).fused_spec() is None       # 600 values
    assert LinearPolicy(env.spec, FeatureStack([RandFourierFeat(5, 10, 1.)])).fused_spec() is None               # 5 != 6 inputs
    assert LinearPolicy(env.spec, FeatureStack([lambda x: x])).fused_spec() is None
    assert pol(to.zeros(6, dtype=to.float64)).shape == (1,) and pol(to.zeros(4, 6, dtype=to.float64)).shape == (4, 1)


# ------------------------------------------------------------------------------------------ return statistics
@pytest.mark.parametrize('epsilon', [1, 0.5, 0.1])
@pytest.mark.parametrize('num_ro', [10, 20])
def test_select_cvar(epsilon, num_ro):
    """ Pyrado/tests/test_sampling.py:128-148 """
    import random
    from simurlacra_amd.sampling import select_cvar
    rollouts = [StepSequence(rewards=[i], observations=[i], actions=[i]) for i in range(num_ro)]
    ro_shuf = list(rollouts)
    random.shuffle(ro_shuf)
    ro_cv = select_cvar(ro_shuf, epsilon, 1)
    cv = sum(map(lambda ro: ro.discounted_return(1), ro_cv))/len(ro_cv)
    nq = int(num_ro*epsilon)
    assert cv == sum(range(nq))/nq


def test_bootstrapping():
    """ Pyrado/tests/test_sampling.py:236-285: same resampling call sequence as the reference, so the interval can be
    recomputed by hand from NumPy's generator """
    from simurlacra_amd.sampling import bootstrap_ci
    sample = np.array([30, 37, 36, 43, 42, 43, 43, 46, 41, 42])
    m, ci = bootstrap_ci(sample, np.mean, num_reps=20, alpha=0.1, ci_sides=2, seed=123)
    np.random.seed(123)
    means = np.random.choice(sample, (sample.shape[0], 20), replace=True).mean(axis=0)
    d_lo, d_up = np.percentile(means - sample.mean(), [5., 95.])
    assert m == sample.mean() and ci == pytest.approx([sample.mean() - d_up, sample.mean() - d_lo])
    x = np.random.default_rng(0).normal(10, 1, 40)
    m, ci2 = bootstrap_ci(x, np.mean, num_reps=1000, alpha=0.05, ci_sides=2, seed=1)
    assert ci2[0] < x.mean() < ci2[1] and 0.3 < ci2[1] - ci2[0] < 0.9          # about +-1.96/sqrt(40)
    mb, _ = bootstrap_ci(x, np.mean, num_reps=1000, alpha=0.05, ci_sides=2, bias_correction=True, seed=1)
    assert mb != m and abs(mb - m) < 0.05
    _, ci1 = bootstrap_ci(x, np.mean, num_reps=768, alpha=0.05, ci_sides=1, seed=1)
    assert ci1[0] == -np.inf and x.mean() < ci1[1] < ci2[1] + 0.1
    _, cit = bootstrap_ci(x, np.mean, num_reps=768, alpha=0.05, ci_sides=2, studentized=True, seed=1)
    assert cit[0] < x.mean() < cit[1]
    with pytest.warns(UserWarning):
        assert bootstrap_ci(np.ones(8), np.mean, num_reps=10, alpha=0.1, ci_sides=2, studentized=True)[1] == [-np.inf, np.inf]


def test_downsampling_and_running_norm_wrappers():
    """ DownsamplingWrapper (Pyrado/pyrado/environment_wrappers/downsampling.py:93-124) and ObsRunningNormWrapper
    (observation_normalization.py:109-135) on the host: held actions, the observation window that outlives a reset, the
    'downsampling' domain parameter, the running bounds incl. the widening of collapsed ones; the batched engine
    refuses both (their state spans rollouts). """
    import torch
    from simurlacra_amd.environment_wrappers import DownsamplingWrapper, ObsRunningNormWrapper
    from simurlacra_amd.sampling.batch import EnvPlan
    from simurlacra_amd.utils.normalizing import RunningNormalizer
    sim = OneMassOscillatorSim(dt=0.01, max_steps=20)
    ds = DownsamplingWrapper(sim, factor=3)
    acts = [ds._process_act(np.array([float(k)])) for k in range(7)]
    assert [float(a[0]) for a in acts] == [0., 0., 0., 3., 3., 3., 6.]
    obs = [ds._process_obs(np.array([float(k), 1.])) for k in range(5)]
    assert obs[0].tolist() == [0., 1.] and obs[1].tolist() == [0.5, 1.] and obs[4].tolist() == [3., 1.]   # mean of the last 3
    dp = {}
    ds._save_domain_param(dp)
    assert dp['downsampling'] == 3 and ds.domain_param['downsampling'] == 3
    ds._load_domain_param(dict(downsampling=np.array(2.)))
    assert ds.factor == 2
    ds.factor = 4
    assert [float(ds._process_act(np.array([float(k)]))[0]) for k in range(5)] == [0., 0., 0., 0., 4.]
    assert np.allclose(ds._process_obs(np.array([9., 1.])), [(3 + 4 + 9)/3., 1.])    # the window kept its first length
    seeded = DownsamplingWrapper(sim, factor=2, obs_filter_fcn=lambda w: np.max(w, axis=0), init_obs=np.array([5., -5.]))
    assert seeded._process_obs(np.array([1., 1.])).tolist() == [5., 1.]
    with pytest.raises(NotImplementedError):
        EnvPlan(ds)
    norm = RunningNormalizer()
    norm(np.array([0., 10.]))
    out = norm(np.array([2., 20.]))
    assert out.tolist() == [1., 1.]
    # the first sample widened the collapsed bounds by eps/2 = 5e-4 on both sides, and that lower bound stays
    assert np.allclose(norm(np.array([1., 15.])), [1.0005/2.0005*2 - 1, 5.0005/10.0005*2 - 1], rtol=1e-12)
    with pytest.raises(Exception):
        norm(np.array([1., 2., 3.]))
    tn = RunningNormalizer()
    tn(torch.tensor([[0., 1.], [4., 3.]], dtype=torch.float64))
    assert torch.equal(tn(torch.tensor([[2., 2.]], dtype=torch.float64)), torch.tensor([[0., 0.]], dtype=torch.float64))
    rn = ObsRunningNormWrapper(sim)
    assert rn.obs_space.bound_up.tolist() == [1., 1.] and rn.obs_space.bound_lo.tolist() == [-1., -1.]
    assert np.allclose(rn._process_obs(np.array([3., 4.])), 0., atol=1e-9)   # one sample: the middle of its widened bounds
    with pytest.raises(NotImplementedError):
        EnvPlan(rn)
