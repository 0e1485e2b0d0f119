"""
CPU tests that hold both oracles to hand-computed one-step numbers (tests/anchors.py) for the simulations whose
dynamics the reference's own tests never pin: QCartPole, BallOnBeam, QBallBalancer (full and simplified dynamics),
Pendulum, and the QBallBalancer inverse kinematics (first two optimizer steps by hand, all 100 through
torch.optim.SGD + autograd, the libraries the reference itself calls).
"""
import numpy as np
import pytest

from oracle import oracle_lib as orc, pysim_oracle_np as onp
from simurlacra_amd import _abi as abi, envdefs
from tests import anchors

RTOL = 1e-13

STEP_ANCHORS = [
    ('qcp-su', abi.ENV_QCP_SU, anchors.qcartpole_step, {}),
    ('bob', abi.ENV_BOB, anchors.ball_on_beam_step, {}),
    ('pend', abi.ENV_PEND, anchors.pendulum_step, {}),
    ('qbb', abi.ENV_QBB, lambda: anchors.qbb_step(False), {}),
    ('qbb', abi.ENV_QBB, lambda: anchors.qbb_step(True), dict(simplified=True)),
]
IDS = ['qcp', 'bob', 'pend', 'qbb-full', 'qbb-simplified']


@pytest.mark.parametrize('name,kind,make,opt', STEP_ANCHORS, ids=IDS)
def test_c_oracle_one_step_matches_hand_computed(name, kind, make, opt):
    a = make()
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    dp = np.array(a['dp'])[:, None]
    task_c = envdefs.default_task_spec(kind).to_c(S, A)
    flags = abi.F_SIMPLIFIED_DYN if opt.get('simplified') else 0
    mem = orc.reset(kind, a['dt'], 100, task_c, np.array(a['state'])[:, None], dp, flags=flags)
    if 'plate' in a:
        mem['state'][S:S + 2, 0] = a['plate']   # the hidden plate angles, given instead of solved for
    obs, rew, done, st, act_after = orc.step(kind, a['dt'], 100, task_c, mem, np.array(a['act'])[:, None], dp, flags=flags)
    np.testing.assert_allclose(mem['state'][:S, 0], a['next'], rtol=RTOL, atol=0)
    np.testing.assert_allclose(obs[:, 0], a['obs'], rtol=RTOL, atol=0)
    assert rew[0] == pytest.approx(a['rew'], rel=RTOL)
    assert done[0] == 0 and st[0] == 0
    if 'plate_next' in a:
        np.testing.assert_allclose(mem['state'][S:S + 2, 0], a['plate_next'], rtol=RTOL, atol=0)
        np.testing.assert_array_equal(act_after[:, 0], a['act'] if opt.get('simplified') else a['act_after'])


@pytest.mark.parametrize('name,kind,make,opt', STEP_ANCHORS, ids=IDS)
def test_numpy_oracle_one_step_matches_hand_computed(name, kind, make, opt):
    a = make()
    env = onp.NpEnv(name, a['dt'], 100, simplified_dyn=bool(opt.get('simplified')))
    env.reset(np.array(a['state']), dict(zip(onp.dp_keys(name), a['dp'])))
    if 'plate' in a:
        env.plate_angs = np.array(a['plate'])
    act = np.array(a['act'])
    obs, rew, done = env.step(act)[:3]
    np.testing.assert_allclose(env.state, a['next'], rtol=RTOL, atol=0)
    np.testing.assert_allclose(obs, a['obs'], rtol=RTOL, atol=0)
    assert rew == pytest.approx(a['rew'], rel=RTOL) and not done
    if 'plate_next' in a:
        np.testing.assert_allclose(env.plate_angs, a['plate_next'], rtol=RTOL, atol=0)


@pytest.mark.parametrize('th', [-1.2, -0.3, 0.0, 0.45, 1.4])
def test_qbb_ik_first_two_optimizer_steps_by_hand(th):
    r, l_plate = 0.0254, 0.275
    tip1, tip2 = anchors.qbb_ik_tips(th, r, l_plate)
    np.testing.assert_allclose(orc.qbb_ik_n(th, r, l_plate, 1)[1], tip1, rtol=1e-14)
    np.testing.assert_allclose(orc.qbb_ik_n(th, r, l_plate, 2)[1], tip2, rtol=1e-14)


@pytest.mark.parametrize('th,r,l_plate', [(-1.4, 0.0254, 0.275), (-0.5, 0.03, 0.25), (0., 0.0254, 0.275),
                                         (0.3, 0.02, 0.3), (1.45, 0.0254, 0.275)])
def test_qbb_ik_matches_torch_sgd_with_autograd(th, r, l_plate):
    want = anchors.qbb_ik_torch(th, r, l_plate)
    assert orc.qbb_ik(th, r, l_plate) == pytest.approx(want, abs=1e-13)
    assert onp.qbb_ik(th, r, l_plate) == pytest.approx(want, abs=1e-13)
