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
from .initialization import init_param as init_module


class LinearPolicy(Policy):
    """ One weight per (action, feature value) pair; the FeatureStack turns an observation into the feature values.  The
    weights live in a bias-free nn.Linear called `net`, as callers of the reference expect (:33). """
    name: str = 'lin'

    def __init__(self, spec: EnvSpec, feats: FeatureStack, init_param_kwargs: dict = None, use_cuda: bool = False):
        if not isinstance(feats, FeatureStack):
            raise TypeErr(given=feats, expected_type=FeatureStack)
        super().__init__(spec, use_cuda)
        self._feats = feats
        self._num_obs, self._num_act = (sp.flat_dim for sp in (spec.obs_space, spec.act_space))
        self.num_active_feat = feats.get_num_feat(self._num_obs)
        self.net = nn.Linear(self.num_active_feat, self._num_act, bias=False).double()
        self.init_param(**(init_param_kwargs or {}))
        self.to(self.device)

    features = property(lambda self: self._feats, doc='the FeatureStack')

    def init_param(self, init_values: to.Tensor = None, **kwargs):
        """ Given values are taken as they are; without them the layer is initialised by the rules of initialization.py. """
        if init_values is not None:
            self.param_values = init_values
            return
        init_module(self.net, **kwargs)

    def eval_feats(self, obs: to.Tensor) -> to.Tensor:
        return self._feats(obs)

    def forward(self, obs: to.Tensor) -> to.Tensor:
        obs = obs.to(self.device)
        act = self.net(self.eval_feats(obs))
        return act.flatten() if obs.ndimension() < 2 else act   # one observation gives one action vector

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
