"""
FNN and FNNPolicy (Pyrado/pyrado/policies/fnn.py:14-203): Linear -> nonlinearity ... -> Linear (-> output nonlinearity).
"""
from typing import Callable, Sequence

import torch as to
import torch.nn as nn
from torch.nn.utils import convert_parameters as cp

from ..specs import PolicySpec
from ..utils.data_types import EnvSpec
from .base import Policy
from .initialization import init_param

_FUSABLE_NL = {None: None, to.tanh: 'tanh', to.relu: 'relu', to.sigmoid: 'sigmoid'}
try:
    import torch.nn.functional as F
    _FUSABLE_NL.update({F.relu: 'relu', F.tanh: 'tanh', F.sigmoid: 'sigmoid'})
except Exception:
    pass


class FNN(nn.Module):
    def __init__(self, input_size: int, output_size: int, hidden_sizes: Sequence[int], hidden_nonlin, dropout: float = 0.,
                 output_nonlin: Callable = None, init_param_kwargs: dict = None, use_cuda: bool = False):
        self._device = 'cuda' if use_cuda and to.cuda.is_available() else 'cpu'
        super().__init__()
        hidden_sizes = list(hidden_sizes)
        self._hidden_nonlin = (list(hidden_nonlin) if isinstance(hidden_nonlin, (list, tuple))
                               else len(hidden_sizes)*[hidden_nonlin])
        self._dropout, self._output_nonlin = dropout, output_nonlin
        self.hidden_layers = nn.ModuleList()
        last = input_size
        for hs in hidden_sizes:
            self.hidden_layers.append(nn.Linear(last, hs))
            last = hs
            if dropout > 0:
                self.hidden_layers.append(nn.Dropout(p=dropout))
        self.output_layer = nn.Linear(last, output_size)
        self.double()
        self.init_param(None, **(init_param_kwargs or dict()))
        self.to(self.device)

    @property
    def device(self) -> str:
        return self._device

    @property
    def dropout(self):
        return self._dropout

    @property
    def hidden_nonlin(self):
        return self._hidden_nonlin

    @property
    def output_nonlin(self):
        return self._output_nonlin

    @property
    def param_values(self) -> to.Tensor:
        return cp.parameters_to_vector(self.parameters())

    @param_values.setter
    def param_values(self, param):
        cp.vector_to_parameters(param, self.parameters())

    def init_param(self, init_values: to.Tensor = None, **kwargs):
        if init_values is None:
            for layer in self.hidden_layers:
                init_param(layer, **kwargs)
            init_param(self.output_layer, **kwargs)
        else:
            self.param_values = init_values

    def forward(self, input: to.Tensor) -> to.Tensor:
        x = input.to(self.device)
        lin = 0
        for layer in self.hidden_layers:
            x = layer(x)
            if isinstance(layer, nn.Linear):
                nl = self._hidden_nonlin[lin]
                lin += 1
                if nl is not None:
                    x = nl(x)
        x = self.output_layer(x)
        return x if self._output_nonlin is None else self._output_nonlin(x)


class FNNPolicy(Policy):
    name: str = 'fnn'

    def __init__(self, spec: EnvSpec, hidden_sizes: Sequence[int], hidden_nonlin, dropout: float = 0.,
                 output_nonlin: Callable = None, init_param_kwargs: dict = None, use_cuda: bool = False):
        super().__init__(spec, use_cuda)
        self.net = FNN(spec.obs_space.flat_dim, spec.act_space.flat_dim, hidden_sizes, hidden_nonlin, dropout,
                       output_nonlin, init_param_kwargs, use_cuda)

    def init_param(self, init_values: to.Tensor = None, **kwargs):
        if init_values is None:
            self.net.init_param(None, **kwargs)
        else:
            self.param_values = init_values

    def forward(self, obs: to.Tensor) -> to.Tensor:
        return self.net(obs)

    def fused_spec(self):
        net = self.net
        hidden = [l.out_features for l in net.hidden_layers if isinstance(l, nn.Linear)]
        if net.dropout > 0 or not 1 <= len(hidden) <= 4 or max(hidden) > 64:
            return None
        try:
            nls = [_FUSABLE_NL[nl] for nl in net.hidden_nonlin]
            out_nl = _FUSABLE_NL[net.output_nonlin]
        except (KeyError, TypeError):
            return None
        return PolicySpec('fnn', self.env_spec.obs_space.flat_dim, self.env_spec.act_space.flat_dim, hidden=hidden,
                          hidden_nonlin=nls, out_nonlin=out_nl)
