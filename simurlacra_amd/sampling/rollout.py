"""
rollout(): one trajectory of a policy in an environment (Pyrado/pyrado/sampling/rollout.py:27-291).

For the simulations and policies the kernels cover, the whole rollout (reset, T x (policy, step), bookkeeping) is one
launch of the fused kernel with a batch of one; every other combination runs the reference's step loop with
env.reset() / env.step() (each a device call for the supported simulations) and the policy evaluated by torch.
"""
import time

import numpy as np
import torch as to

from .. import inf
from ..environment_wrappers.domain_randomization import (DomainRandWrapperBuffer, DomainRandWrapperLive,
                                                          MetaDomainRandWrapper)
from ..environment_wrappers.base import EnvWrapper
from ..environment_wrappers.utils import all_envs, inner_env
from ..environments.base import Env
from ..environments.sim_base import SimEnv
from ..utils.data_types import RenderMode
from ..utils.exceptions import TypeErr, ValueErr
from .step_sequence import StepSequence


def _fused_plan(env, policy, **conds):
    """ The EnvPlan if this call can run as one fused launch, else None. """
    if any(conds.values()):
        return None
    spec = policy.fused_spec() if hasattr(policy, 'fused_spec') else None
    if spec is None or env.max_steps == inf:
        return None
    try:
        from .batch import EnvPlan
        plan = EnvPlan(env)
    except NotImplementedError:
        return None
    return plan


def _reset_host_side(env, init_state, domain_param):
    """ What env.reset(init_state, domain_param) does on the host before the dynamics start: the wrappers choose the
    domain parameters (environment_wrappers/domain_randomization.py:111-118,179-199), the simulation takes them over
    and rebuilds its spaces and task (pysim/base.py:75-87,119-149), and the init state is drawn from the init space. """
    for e in all_envs(env):
        if isinstance(e, MetaDomainRandWrapper):
            continue
        if isinstance(e, DomainRandWrapperLive) and domain_param is None:
            e.randomizer.randomize(num_samples=1)
            domain_param = e.randomizer.get_params(format='dict', dtype='numpy')
        elif isinstance(e, DomainRandWrapperBuffer) and domain_param is None:
            if isinstance(e.buffer, dict):
                domain_param = e.buffer
            else:
                domain_param = e.buffer[e.ring_idx]
                e._ring_idx = (e.ring_idx + 1) % len(e.buffer)
    sim = inner_env(env)
    sim._curr_step = 0
    if domain_param is not None:
        for e in all_envs(env):  # EnvWrapper.reset: every wrapper takes its own entries (e.g. 'act_delay')
            if isinstance(e, EnvWrapper):
                e._load_domain_param(domain_param)
        sim.domain_param = domain_param
    if init_state is None:
        init_state = sim.init_space.sample_uniform()
    elif not isinstance(init_state, np.ndarray):
        try:
            init_state = np.array(init_state, dtype=np.float64)
        except Exception:
            raise TypeErr(given=init_state, expected_type=[np.ndarray, list])
    sim.task.reset(env_spec=sim.spec)
    return np.asarray(init_state, dtype=np.float64)


def rollout(env: Env, policy, eval: bool = False, max_steps: int = None, reset_kwargs: dict = None,
            render_mode: RenderMode = RenderMode(), render_step: int = 1, bernoulli_reset: float = None,
            no_reset: bool = False, no_close: bool = False, record_dts: bool = False,
            stop_on_done: bool = True) -> StepSequence:
    if not isinstance(env, Env):
        raise TypeErr(given=env, expected_type=Env)
    if not isinstance(eval, bool):
        raise TypeErr(given=eval, expected_type=bool)
    if not (isinstance(reset_kwargs, dict) or reset_kwargs is None):
        raise TypeErr(given=reset_kwargs, expected_type=dict)
    if not isinstance(render_mode, RenderMode):
        raise TypeErr(given=render_mode, expected_type=RenderMode)
    if max_steps is not None:
        env.max_steps = max_steps
    reset_kwargs = dict() if reset_kwargs is None else reset_kwargs

    plan = _fused_plan(env, policy, bernoulli=False, no_reset=no_reset, dts=record_dts,
                       render=render_mode.text or render_mode.video, recurrent=getattr(policy, 'is_recurrent', False))
    if plan is not None:
        return _rollout_fused(env, plan, policy, eval, reset_kwargs, stop_on_done, no_close, bernoulli_reset)
    return _rollout_stepwise(env, policy, eval, reset_kwargs, render_mode, render_step, bernoulli_reset, no_reset,
                             no_close, record_dts, stop_on_done)


def _rollout_fused(env, plan, policy, eval, reset_kwargs, stop_on_done, no_close, bernoulli_reset=None):
    from ..engine import get_engine
    from .batch import RolloutBatch, run_batch
    init_state = _reset_host_side(env, reset_kwargs.get('init_state'), reset_kwargs.get('domain_param'))
    if hasattr(policy, 'reset'):
        policy.reset()
        policy.eval() if eval else policy.train()
    eng = get_engine()
    sim = plan.sim
    dp = eng.f64(plan.dp_column().reshape(-1, 1))
    # the noise wrappers of the reference draw from NumPy's global generator: take this rollout's key from it
    spec = policy.fused_spec()
    stochastic = plan.wrappers or spec.expl_kind is not None or bernoulli_reset is not None
    noise_seed = int(np.random.randint(0, 2**31 - 1)) if stochastic else 0
    out = run_batch(plan, eng, policy, eng.f64(init_state.reshape(-1, 1)), dp, stop_on_done=stop_on_done, record=True,
                    noise_seed=noise_seed, bernoulli_reset=bernoulli_reset)
    batch = RolloutBatch(plan, out, dp)
    batch.check()
    ro = batch[0]
    ro.rollout_info['domain_param'] = env.domain_param
    sim.state = out['final_state'][:, 0].cpu().numpy()
    sim._curr_step = int(out['length'][0].item())
    if not no_close:
        env.close()
    return ro


def _rollout_stepwise(env, policy, eval, reset_kwargs, render_mode, render_step, bernoulli_reset, no_reset, no_close,
                      record_dts, stop_on_done):
    obs_hist, act_hist, rew_hist, info_hist = [], [], [], []
    dts = ([], [], [])
    recurrent = getattr(policy, 'is_recurrent', False)
    hidden_hist = []
    obs = env.reset(**reset_kwargs) if not no_reset else np.zeros(env.obs_space.shape)
    if isinstance(policy, to.nn.Module):
        if hasattr(policy, 'reset'):
            policy.reset()
        policy.eval() if eval else policy.train()
    hidden = policy.init_hidden() if recurrent else None
    rollout_info = dict(env_spec=env.spec)
    if isinstance(inner_env(env), SimEnv):
        rollout_info['domain_param'] = env.domain_param
    done = False
    t_post_step = time.time()
    while not (done and stop_on_done) and env.curr_step < env.max_steps:
        t_start = time.time()
        if np.isnan(obs).any():
            raise ValueErr(msg=f'At least one observation value is NaN! {obs}')
        obs_to = to.from_numpy(np.asarray(obs)).type(to.float64)
        with to.no_grad():
            if recurrent:
                act_to, hidden_next = policy(obs_to, hidden)
            else:
                act_to = policy(obs_to)
        act = act_to.detach().cpu().numpy().astype(np.float64)
        if np.isnan(act).any():
            raise ValueErr(msg=f'At least one action value is NaN! {act}')
        t_post_policy = time.time()
        obs_next, rew, done, env_info = env.step(act)
        dts[0].append(t_post_policy - t_start)
        dts[1].append(time.time() - t_post_policy)
        dts[2].append(t_start - t_post_step)
        t_post_step = time.time()
        obs_hist.append(obs)
        act_hist.append(act)
        rew_hist.append(rew)
        info_hist.append(env_info)
        if recurrent:
            hidden_hist.append(hidden)
            hidden = hidden_next
        obs = obs_next
        env.render(render_mode, render_step)
        if bernoulli_reset is not None:
            assert 0. <= bernoulli_reset <= 1.
            if np.random.binomial(1, bernoulli_reset):
                break
    if not no_close:
        env.close()
    obs_hist.append(obs)
    res = StepSequence(observations=obs_hist, actions=act_hist, rewards=rew_hist, rollout_info=rollout_info,
                       env_infos=info_hist, complete=True)
    if recurrent:
        res.add_data('hidden_states', hidden_hist)
    if record_dts:
        res.add_data('dts_policy', dts[0])
        res.add_data('dts_step', dts[1])
        res.add_data('dts_remainder', dts[2])
    return res
