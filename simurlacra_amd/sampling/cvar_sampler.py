"""
select_cvar / CVaRSampler (Pyrado/pyrado/sampling/cvar_sampler.py:9-107): keep the epsilon-quantile of rollouts with the
lowest (discounted) return, so that their mean return is the CVaR(epsilon) of the sampled set.

For a RolloutBatch the returns are ranked on the device (one argsort over N doubles) and only the kept rollouts are
ever turned into StepSequence objects.
"""
from typing import List

import numpy as np
import torch

from .batch import RolloutBatch
from .parallel_sampler import SamplerBase


def _discounted_returns(batch: RolloutBatch, gamma: float) -> torch.Tensor:
    out = batch.device_results
    lo, n = batch._offset, len(batch)
    if gamma == 1.:
        return out['ret'][lo:lo + n]
    if 'rew' not in out:
        raise ValueError('discounted returns need recorded trajectories')
    rew = torch.nan_to_num(out['rew'][:, lo:lo + n], nan=0.)   # steps past the end of a rollout hold NaN
    disc = gamma**torch.arange(rew.shape[0], dtype=torch.float64, device=rew.device)
    return (disc.unsqueeze(1)*rew).sum(dim=0)


def select_cvar(rollouts, epsilon: float, gamma: float = 1.):
    """ The floor(len * epsilon) rollouts with the lowest discounted return, in ascending order.  A plain list is
    sorted in place, as the reference does. """
    if isinstance(rollouts, RolloutBatch):
        n = int(np.floor(len(rollouts)*epsilon))
        order = torch.argsort(_discounted_returns(rollouts, gamma), stable=True)[:n].cpu().numpy()
        return [rollouts[int(i)] for i in order]
    if not isinstance(rollouts, list):   # e.g. the concatenation of several batches
        rollouts = list(rollouts)
    rollouts.sort(key=lambda r: r.discounted_return(gamma))
    return rollouts[0:int(np.floor(len(rollouts)*epsilon))]


class CVaRSampler(SamplerBase):
    """ Samples 1/epsilon times as many rollouts with the wrapped sampler and returns their epsilon-quantile. """

    def __init__(self, wrapped_sampler, epsilon: float, gamma: float = 1., *, min_rollouts: int = None,
                 min_steps: int = None):
        self._wrapped_sampler = wrapped_sampler
        self.epsilon = epsilon
        self.gamma = gamma
        super().__init__(min_rollouts=min_rollouts, min_steps=min_steps)
        self.set_min_count(min_rollouts=min_rollouts, min_steps=min_steps)
        self.full_set_stats = {}

    def set_min_count(self, min_rollouts=None, min_steps=None):
        super().set_min_count(min_rollouts=min_rollouts, min_steps=min_steps)
        # the (1 - epsilon) quantile is discarded, so the full set has to be larger by 1/epsilon
        if min_rollouts is not None:
            min_rollouts = int(min_rollouts/self.epsilon)
        if min_steps is not None:
            min_steps = int(min_steps/self.epsilon)
        self._wrapped_sampler.set_min_count(min_rollouts=min_rollouts, min_steps=min_steps)

    def reinit(self, env=None, policy=None):
        self._wrapped_sampler.reinit(env=env, policy=policy)

    def sample(self) -> List:
        fullset = self._wrapped_sampler.sample()
        rets = fullset.undiscounted_returns()
        # what the reference logs about the full set
        self.full_set_stats = {'full num rollouts': len(fullset), 'full avg rollout len': float(np.mean(fullset.lengths)),
                               'full avg return': float(np.mean(rets)), 'full median return': float(np.median(rets)),
                               'full std return': float(np.std(rets))}
        return select_cvar(fullset, self.epsilon, self.gamma)
