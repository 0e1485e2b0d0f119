"""
ParallelSampler (later upstream name: ParallelRolloutSampler), Pyrado/pyrado/sampling/parallel_sampler.py:44-123.

Same constructor and sample() contract: `min_rollouts` -> exactly that many rollouts, in order; `min_steps` -> at
least that many steps in total, nothing dropped.  The worker processes of the reference are replaced by GPU lanes:
`num_envs` is accepted for compatibility and ignored.  Under torch.distributed (one process per GPU) the rollouts are
split over the ranks and the per-rollout returns, lengths and domain parameters are all-gathered.
"""
import math

import torch

from .. import inf
from ..utils.exceptions import ValueErr
from . import distributed as D
from .batch import EnvPlan, GatheredBatch, RolloutBatch, run_batch, sample_domain_params, sample_init_states


class SamplerBase:
    def __init__(self, *, min_rollouts: int = None, min_steps: int = None):
        if min_rollouts is None and min_steps is None:
            raise ValueErr(msg='At least one of min_rollouts or min_steps must be given')
        self.min_rollouts, self.min_steps = min_rollouts, min_steps

    def set_min_count(self, min_rollouts: int = None, min_steps: int = None):
        """ SamplerBase.set_min_count (Pyrado/pyrado/sampling/sampler.py): change the amount to sample """
        if min_rollouts is None and min_steps is None:
            raise ValueErr(msg='At least one of min_rollouts or min_steps must be given')
        self.min_rollouts, self.min_steps = min_rollouts, min_steps


class ParallelSampler(SamplerBase):
    def __init__(self, env, policy, num_envs: int = 1, *, min_rollouts: int = None, min_steps: int = None,
                 bernoulli_reset: float = None, seed: int = None, record_trajectories: bool = True):
        super().__init__(min_rollouts=min_rollouts, min_steps=min_steps)
        if bernoulli_reset is not None and not 0. <= bernoulli_reset <= 1.:
            raise ValueErr(given=bernoulli_reset, ge_constraint='0', le_constraint='1')
        self.env, self.policy, self.num_envs = env, policy, num_envs
        self.bernoulli_reset = bernoulli_reset
        self.record_trajectories = record_trajectories
        self.seed = 0 if seed is None else int(seed)
        self._calls = 0
        self._plan = EnvPlan(env)

    def reinit(self, env=None, policy=None, bernoulli_reset=None):
        if env is not None:
            self.env = env
            self._plan = EnvPlan(env)
        if policy is not None:
            self.policy = policy
        if bernoulli_reset is not None:
            self.bernoulli_reset = bernoulli_reset

    def set_seed(self, seed: int):
        self.seed, self._calls = int(seed), 0

    def _sample_n(self, eng, n: int, global_offset: int, n_global: int):
        """ This rank's share of n_global rollouts: global env indices [global_offset + start, global_offset + stop) """
        rank, ws = D.world()
        start, stop = D.shard_range(n_global, rank, ws)
        m = stop - start
        plan = self._plan
        key = self.seed + 7919*self._calls
        dp, dp_index = sample_domain_params(plan, eng, m, seed=key, env_offset=global_offset + start,
                                           n_global=n_global, shard_start=start)
        init = sample_init_states(plan, eng, m, dp, dp_index, seed=key, env_offset=global_offset + start)
        out = run_batch(plan, eng, self.policy, init, dp, dp_index, record=self.record_trajectories, noise_seed=key,
                        env_offset=global_offset + start, bernoulli_reset=self.bernoulli_reset)
        return RolloutBatch(plan, out, dp, dp_index)

    def sample(self) -> RolloutBatch:
        from ..engine import get_engine
        eng = get_engine()
        T = self.env.max_steps
        if T == inf:
            raise ValueErr(msg='ParallelSampler needs an environment with a finite max_steps')
        if self.min_steps is None:
            batch = self._sample_n(eng, self.min_rollouts, 0, self.min_rollouts)
            self._calls += 1
            batch = self._gathered(batch, self.min_rollouts)
            batch.check()          # after the gather: every rank raises for a NaN on any of them
            return batch
        # at least min_steps steps (and min_rollouts rollouts): top up until both are met, keep everything
        n = max(self.min_rollouts or 0, math.ceil(self.min_steps/int(T)))
        parts, steps, done_rollouts = [], 0, 0
        while steps < self.min_steps or done_rollouts < (self.min_rollouts or 0):
            b = self._gathered(self._sample_n(eng, n, done_rollouts, n), n)
            b.check()
            parts.append(b)
            steps += b.num_steps
            done_rollouts += n
            self._calls += 1
            remaining = self.min_steps - steps
            if remaining > 0:
                mean_len = max(1., steps/done_rollouts)
                n = max(1, math.ceil(remaining/mean_len))
        return parts[0] if len(parts) == 1 else _ConcatBatches(parts)

    def _gathered(self, batch: RolloutBatch, n_global: int):
        """ Under torch.distributed: all-gather the per-rollout results (returns, lengths, status, domain parameters)
        and return the view of all n_global rollouts that is the same on every rank. """
        rank, ws = D.world()
        if ws == 1:
            return batch
        out = batch.device_results
        g = dict(ret=D.all_gather_rows(out['ret'], n_global), length=D.all_gather_rows(out['length'], n_global),
                 status=D.all_gather_rows(out['status'], n_global))
        if batch._dp_index is None:
            g['domain_param'] = D.all_gather_rows(batch._dp.t().contiguous(), n_global).t()
        return GatheredBatch(batch, g, D.shard_range(n_global, rank, ws)[0], n_global)


ParallelRolloutSampler = ParallelSampler


class _ConcatBatches:
    """ Several RolloutBatch parts behaving as one list of rollouts (min_steps top-up). """

    def __init__(self, parts):
        self.parts = parts

    def __len__(self):
        return sum(len(p) for p in self.parts)

    def __getitem__(self, i):
        i = int(i)
        if i < 0:
            i += len(self)
        for p in self.parts:
            if i < len(p):
                return p[i]
            i -= len(p)
        raise IndexError

    def __iter__(self):
        for p in self.parts:
            yield from p

    def undiscounted_returns(self):
        import numpy as np
        return np.concatenate([p.undiscounted_returns() for p in self.parts])

    @property
    def lengths(self):
        import numpy as np
        return np.concatenate([p.lengths for p in self.parts])

    @property
    def num_steps(self):
        return int(sum(p.num_steps for p in self.parts))
