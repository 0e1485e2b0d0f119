"""
ParameterExplorationSampler (Pyrado/pyrado/sampling/parameter_exploration_sampler.py:23-201): every policy-parameter
set of a population is evaluated on the SAME `num_rollouts_per_param` (domain parameter, initial state) pairs
(common random numbers); results are grouped per parameter set.

All pop x R rollouts are one launch of the fused kernel: lane j*R + r runs parameter set j on pair r
(policy_index = j, dp_index = r); the per-set mean returns are reduced on the device.
"""
from typing import List, NamedTuple, Sequence

import numpy as np
import torch as to

from ..environment_wrappers.domain_randomization import DomainRandWrapper, remove_all_dr_wrappers
from ..environment_wrappers.utils import typed_env
from . import distributed as D
from .. import _abi as abi
from ..utils.exceptions import ValueErr
from .batch import EnvPlan, RolloutBatch, SummaryRollouts, run_batch, sample_domain_params, sample_init_states


class ParameterSample(NamedTuple):
    params: to.Tensor
    rollouts: Sequence

    @property
    def mean_undiscounted_return(self):
        if hasattr(self.rollouts, 'undiscounted_returns'):
            return float(np.mean(self.rollouts.undiscounted_returns()))
        return float(np.mean([r.undiscounted_return() for r in self.rollouts]))

    @property
    def num_rollouts(self):
        return len(self.rollouts)


class _LazySamples(Sequence):
    """ The ParameterSample objects of a population, built on access: a CEM generation of 4 096 parameter sets reads the
    mean returns and a few hundred elites, not 4 096 Python objects. """

    def __init__(self, n: int, make):
        self._n, self._make, self._cache = n, make, {}

    def __len__(self):
        return self._n

    def __getitem__(self, j):
        if isinstance(j, slice):
            return [self[k] for k in range(*j.indices(self._n))]
        j = int(j)
        if j < 0:
            j += self._n
        if not 0 <= j < self._n:
            raise IndexError(j)
        if j not in self._cache:
            self._cache[j] = self._make(j)
        return self._cache[j]


class ParameterSamplingResult(Sequence):
    def __init__(self, samples: Sequence, mean_returns: np.ndarray = None, mean_returns_device=None,
                 param_sets: to.Tensor = None, num_rollouts: int = None):
        self._samples = samples
        self._mean_returns = mean_returns
        self._means_dev = mean_returns_device     # the same means, still on the GPU (all parameter sets, every rank)
        self._param_sets, self._num_rollouts = param_sets, num_rollouts

    def __getitem__(self, idx):
        res = self._samples[idx]
        if not isinstance(res, ParameterSample):
            mr = None if self._mean_returns is None else self._mean_returns[idx]
            return ParameterSamplingResult(res, mr)
        return res

    def __len__(self):
        return len(self._samples)

    @property
    def parameters(self) -> to.Tensor:
        if self._param_sets is not None:
            return self._param_sets
        return to.stack([s.params for s in self._samples])

    @property
    def mean_returns(self) -> np.ndarray:
        if self._mean_returns is None:
            self._mean_returns = np.array([s.mean_undiscounted_return for s in self._samples])
        return self._mean_returns

    def elites(self, num: int):
        """ The `num` parameter sets with the highest mean return, best first: (indices, their mean returns, their
        parameters [num, num_param]) as CEM.update selects its importance samples
        (Pyrado/pyrado/algorithms/cem.py:116-120).  Ranked on the GPU when the means are still there. """
        num = min(int(num), len(self))
        if self._means_dev is not None:
            from ..engine import get_engine
            idx, val = get_engine().topk(self._means_dev, num)
            idx, val = idx.cpu().numpy().astype(np.int64), val.cpu().numpy()
        else:
            means = np.asarray(self.mean_returns, dtype=np.float64)
            key = np.where(np.isnan(means), np.inf, means)
            idx = np.lexsort((np.arange(len(key)), -key))[:num]
            val = means[idx]
        return idx, val, self.parameters[to.as_tensor(idx)]

    @property
    def rollouts(self) -> list:
        return [s.rollouts for s in self._samples]

    @property
    def num_rollouts(self) -> int:
        if self._num_rollouts is not None:
            return self._num_rollouts
        return int(np.sum([s.num_rollouts for s in self._samples]))


class ParameterExplorationSampler:
    def __init__(self, env, policy, num_envs: int = 1, num_rollouts_per_param: int = 1, seed: int = None,
                 record_trajectories: bool = True):
        self._dp_wrapper = typed_env(env, DomainRandWrapper)
        self._full_plan = EnvPlan(env)               # with the randomization wrappers: source of domain parameters
        if self._dp_wrapper is not None:
            env = remove_all_dr_wrappers(env)
        self.env, self.policy = env, policy
        self._plan = EnvPlan(env)
        self._plan.dr_wrapper = self._full_plan.dr_wrapper
        self.num_rollouts_per_param = num_rollouts_per_param
        self.record_trajectories = record_trajectories
        self.seed = 0 if seed is None else int(seed)
        self._calls = 0

    def sample(self, param_sets: to.Tensor) -> ParameterSamplingResult:
        from ..engine import get_engine
        eng = get_engine()
        plan, R = self._plan, self.num_rollouts_per_param
        param_sets = to.as_tensor(param_sets, dtype=to.float64)
        pop = param_sets.shape[0]
        key = self.seed + 7919*self._calls
        self._calls += 1
        # R (domain parameter, init state) pairs shared by the whole population
        dp, dp_index = sample_domain_params(plan, eng, R, seed=key)
        if dp_index is not None:                     # buffer / fixed parameters: materialise the R columns
            dp = dp[:, dp_index.long()].contiguous()
        init_r = sample_init_states(plan, eng, R, dp, None, seed=key)
        rank, ws = D.world()
        p0, p1 = D.shard_range(pop, rank, ws)        # shard by parameter set: a set's R rollouts stay on one GPU
        my = p1 - p0
        lanes = my*R
        dev = eng.device
        r_idx = to.arange(R, dtype=to.int32, device=dev).repeat(my)
        init = init_r[:, r_idx.long()].contiguous()
        spec = self.policy.fused_spec() if hasattr(self.policy, 'fused_spec') else None
        if spec is None:
            raise NotImplementedError('ParameterExplorationSampler needs a LinearPolicy or FNNPolicy that the rollout '
                                      'kernels can fuse')
        my_params = param_sets[p0:p1].to(dev)
        # one launch for the whole population: lane j*R + r runs parameter set j on the shared pair r
        out = run_batch(plan, eng, self.policy, init, dp, r_idx, record=self.record_trajectories, params=my_params,
                        policy_group=R if my > 1 else 0, noise_seed=key, env_offset=p0*R)
        batch = RolloutBatch(plan, out, dp, r_idx)
        means = eng.segment_mean(out['ret'], R) if lanes > 0 else to.empty(0, dtype=to.float64, device=dev)
        means_dev = D.all_gather_rows(means, pop)
        means = means_dev.cpu().numpy()
        if ws == 1:
            batch.check()
            samples = _LazySamples(pop, lambda j: ParameterSample(params=param_sets[j], rollouts=batch.slice(j*R, R)))
            return ParameterSamplingResult(samples, means, means_dev, param_sets, pop*R)
        # every rank gets the returns, lengths and status words of ALL parameter sets ([pop, R] each; trajectories
        # stay where they were produced), so num_rollouts, len(rollout) and slicing of the result agree on all ranks
        ret = D.all_gather_rows(out['ret'].view(my, R), pop).cpu().numpy()
        length = D.all_gather_rows(out['length'].view(my, R), pop).cpu().numpy()
        status = D.all_gather_rows(out['status'].view(my, R), pop).cpu().numpy()
        if (status & (abi.ST_NAN | abi.ST_BAD_INDEX)).any():
            j, r = np.argwhere(status & (abi.ST_NAN | abi.ST_BAD_INDEX))[0]
            raise ValueErr(msg=f'At least one value is NaN! (rollout {int(r)} of parameter set {int(j)})')
        dp_host = dp.detach().cpu().numpy()          # the R shared columns, identical on every rank

        def dp_of(r):
            return plan.dp_dicts(dp_host, [r])[0]
        def make(j):
            if p0 <= j < p1:
                ros = batch.slice((j - p0)*R, R)
            else:
                ros = SummaryRollouts(ret[j], length[j], dp_of, batch._env_spec)
            return ParameterSample(params=param_sets[j], rollouts=ros)
        return ParameterSamplingResult(_LazySamples(pop, make), means, means_dev, param_sets, pop*R)
