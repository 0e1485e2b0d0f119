"""
Action-noise exploration strategies (Pyrado/pyrado/exploration/stochastic_action.py:17-176): a policy whose action is
a sample from a distribution centred at the wrapped policy's output.

For the policies the rollout kernels can fuse, the sampling is fused as well (psh_policy.expl_*): the kernels add the
noise right after the policy forward, so the exploratory action is what gets recorded and applied — exactly what
rollout() sees when it is handed the exploration strategy as its policy.
"""
from abc import ABC
from dataclasses import replace

import torch as to
from torch.distributions import Distribution

from ..policies.base import Policy
from .normal_noise import DiagNormalNoise
from .uniform_noise import UniformNoise


class StochasticActionExplStrat(Policy, ABC):
    def __init__(self, policy: Policy):
        super().__init__(policy.env_spec)
        self.policy = policy

    @property
    def is_recurrent(self) -> bool:
        return self.policy.is_recurrent

    def init_param(self, init_values: to.Tensor = None, **kwargs):
        self.policy.init_param(init_values, **kwargs)

    def reset(self):
        self.policy.reset()

    def forward(self, obs: to.Tensor, *extra) -> to.Tensor:
        act = self.policy(obs, *extra)
        return self.action_dist_at(act).rsample()  # reparametrised sample around the policy's output

    def evaluate(self, rollout, hidden_states_name: str = 'hidden_states') -> Distribution:
        self.policy.eval()
        acts = self.policy.evaluate(rollout, hidden_states_name)
        return self.action_dist_at(acts)

    def action_dist_at(self, policy_output: to.Tensor) -> Distribution:
        raise NotImplementedError

    # ---- fused rollouts
    def _expl_fields(self) -> dict:
        raise NotImplementedError

    def fused_spec(self):
        spec = self.policy.fused_spec()
        return None if spec is None else replace(spec, **self._expl_fields())

    def fused_params(self) -> to.Tensor:
        return self.policy.fused_params()


class NormalActNoiseExplStrat(StochasticActionExplStrat):
    def __init__(self, policy: Policy, std_init, std_min=0.01, train_mean: bool = False, learnable: bool = True):
        super().__init__(policy)
        self._noise = DiagNormalNoise(noise_dim=policy.env_spec.act_space.flat_dim, std_init=std_init, std_min=std_min,
                                      train_mean=train_mean, learnable=learnable)

    @property
    def noise(self) -> DiagNormalNoise:
        return self._noise

    def action_dist_at(self, policy_output: to.Tensor) -> Distribution:
        return self._noise(policy_output)

    def reset_expl_params(self):
        self._noise.reset_expl_params()

    @property
    def std(self):
        return self._noise.std

    @std.setter
    def std(self, value):
        self._noise.std = value

    @property
    def mean(self):
        return self._noise.mean

    def get_entropy(self):
        return self._noise.get_entropy()

    def _expl_fields(self) -> dict:
        n = self._noise
        return dict(expl_kind='normal', expl_scale=n.std.detach().cpu().numpy(),
                    expl_mean=None if n.mean is None else n.mean.detach().cpu().numpy())


class UniformActNoiseExplStrat(StochasticActionExplStrat):
    def __init__(self, policy: Policy, halfspan_init, halfspan_min=0.01, train_mean: bool = False, learnable: bool = True):
        super().__init__(policy)
        self._noise = UniformNoise(noise_dim=policy.env_spec.act_space.flat_dim, halfspan_init=halfspan_init,
                                   halfspan_min=halfspan_min, train_mean=train_mean, learnable=learnable)

    @property
    def noise(self) -> UniformNoise:
        return self._noise

    def action_dist_at(self, policy_output: to.Tensor) -> Distribution:
        return self._noise(policy_output)

    def reset_expl_params(self):
        self._noise.reset_expl_params()

    @property
    def halfspan(self):
        return self._noise.halfspan

    @halfspan.setter
    def halfspan(self, value):
        self._noise.halfspan = value

    def get_entropy(self):
        return self._noise.get_entropy()

    def _expl_fields(self) -> dict:
        n = self._noise
        return dict(expl_kind='uniform', expl_scale=n.halfspan.detach().cpu().numpy(),
                    expl_mean=None if n.mean is None else n.mean.detach().cpu().numpy())
