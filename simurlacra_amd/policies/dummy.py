"""
DummyPolicy (uniformly random actions) and IdlePolicy (zero actions), Pyrado/pyrado/policies/dummy.py.
"""
import torch as to

from .base import Policy


class IdlePolicy(Policy):
    name: str = 'idle'

    def init_param(self, init_values=None, **kwargs):
        pass

    def forward(self, obs: to.Tensor = None) -> to.Tensor:
        shape = self.env_spec.act_space.shape if obs is None or obs.dim() == 1 else (obs.shape[0],) + self.env_spec.act_space.shape
        return to.zeros(shape, dtype=to.float64)


class DummyPolicy(Policy):
    name: str = 'dummy'

    def init_param(self, init_values=None, **kwargs):
        pass

    def forward(self, obs: to.Tensor = None) -> to.Tensor:
        if obs is not None and obs.dim() == 2:
            return to.stack([to.from_numpy(self.env_spec.act_space.sample_uniform()) for _ in range(obs.shape[0])])
        return to.from_numpy(self.env_spec.act_space.sample_uniform())
