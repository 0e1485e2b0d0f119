"""
Batched policy evaluation over explicit (domain parameter, init state) lists
(Pyrado/pyrado/sampling/parallel_evaluation.py:28-94).  The `pool` argument of the reference functions is accepted
and ignored: the rollouts are lanes of one kernel launch.
"""
import numpy as np

from ..environment_wrappers.domain_randomization import remove_all_dr_wrappers
from .batch import EnvPlan, RolloutBatch, run_batch


def _evaluate(env, policy, dp_dicts, init_states, record=True):
    from ..engine import get_engine
    eng = get_engine()
    plan = EnvPlan(remove_all_dr_wrappers(env))
    n = len(dp_dicts)
    table = np.stack([plan.dp_column(d or {}) for d in dp_dicts], axis=1)
    dp = eng.f64(table)
    if init_states is None or any(s is None for s in init_states):
        init = eng.sample_init(plan.kind, n, dp, seed=0)
    else:
        init = eng.f64(np.stack([np.asarray(s, dtype=np.float64) for s in init_states], axis=1))
    out = run_batch(plan, eng, policy, init, dp, None, record=record, noise_seed=int(np.random.randint(0, 2**31 - 1)))
    batch = RolloutBatch(plan, out, dp)
    batch.check()
    return batch


def eval_domain_params(pool, env, policy, params: list, init_state=None) -> RolloutBatch:
    """ One rollout per domain-parameter dict in `params`. """
    return _evaluate(env, policy, params, None if init_state is None else [init_state]*len(params))


def eval_nominal_domain(pool, env, policy, init_states: list) -> RolloutBatch:
    """ One rollout per initial state with the environment's current (nominal) domain parameters. """
    return _evaluate(env, policy, [None]*len(init_states), init_states)


def eval_randomized_domain(pool, env, randomizer, policy, init_states: list) -> RolloutBatch:
    """ One rollout per initial state, each with freshly drawn domain parameters. """
    randomizer.randomize(len(init_states))
    params = randomizer.get_params(-1, format='list', dtype='numpy')
    params = [params] if isinstance(params, dict) else params
    params = [{k: float(v) for k, v in d.items()} for d in params]
    return _evaluate(env, policy, params, init_states)
