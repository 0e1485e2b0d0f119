"""
LinearPolicy (Pyrado/pyrado/policies/linear.py:11-80): act = W . features(obs), no bias.
"""
import torch as to
import torch.nn as nn

from ..specs import PolicySpec
from ..utils.data_types import EnvSpec
from ..utils.exceptions import TypeErr
from .base import Policy
from .features import FeatureStack
from .initialization import init_param


class LinearPolicy(Policy):
    name: str = 'lin'

    def __init__(self, spec: EnvSpec, feats: FeatureStack, init_param_kwargs: dict = None, use_cuda: bool = False):
        super().__init__(spec, use_cuda)
        if not isinstance(feats, FeatureStack):
            raise TypeErr(given=feats, expected_type=FeatureStack)
        self._num_act, self._num_obs = spec.act_space.flat_dim, spec.obs_space.flat_dim
        self._feats = feats
        self.num_active_feat = feats.get_num_feat(self._num_obs)
        self.net = nn.Linear(self.num_active_feat, self._num_act, bias=False).double()
        self.init_param(None, **(init_param_kwargs or dict()))
        self.to(self.device)

    @property
    def features(self) -> FeatureStack:
        return self._feats

    def init_param(self, init_values: to.Tensor = None, **kwargs):
        if init_values is None:
            init_param(self.net, **kwargs)
        else:
            self.param_values = init_values

    def eval_feats(self, obs: to.Tensor) -> to.Tensor:
        return self._feats(obs)

    def forward(self, obs: to.Tensor) -> to.Tensor:
        obs = obs.to(self.device)
        batched = obs.ndimension() == 2
        act = self.net(self.eval_feats(obs))
        return act if batched else act.flatten()

    def fused_spec(self):
        names = self._feats.fused_names()
        if names is None or not 1 <= len(names) <= 8:
            return None
        spec = PolicySpec('linear', self._num_obs, self._num_act, feats=names)
        try:
            spec.to_c()   # sizes the kernels do not take (too many feature values, mismatching dimensions)
        except ValueError:
            return None
        return spec
