"""
Policy base class (Pyrado/pyrado/policies/base.py:27-156): an nn.Module with a flat parameter-vector view.
Policies are fp64, as the reference makes them by setting torch's default dtype to double.
"""
from abc import ABC, abstractmethod

import torch as to
import torch.nn as nn
import torch.nn.utils.convert_parameters as cp

from ..utils.data_types import EnvSpec
from ..utils.exceptions import ShapeErr, TypeErr


class Policy(nn.Module, ABC):
    """ What the samplers need from a policy: its specification, a flat view of its parameters, forward(); and, for the
    fused rollout kernels, a description of itself (fused_spec). """
    name: str = None

    def __init__(self, spec: EnvSpec, use_cuda: bool = False):
        if not isinstance(spec, EnvSpec):
            raise TypeErr(given=spec, expected_type=EnvSpec)
        nn.Module.__init__(self)
        self._env_spec = spec
        self._device = 'cuda' if (use_cuda and to.cuda.is_available()) else 'cpu'

    env_spec = property(lambda self: self._env_spec)
    device = property(lambda self: self._device)
    is_recurrent = property(lambda self: False)

    # the parameters as one vector (:58-75) ----------------------------------------------------------------------------
    @property
    def num_param(self) -> int:
        return sum(p.data.numel() for p in self.parameters())

    @property
    def param_values(self) -> to.Tensor:
        return cp.parameters_to_vector(self.parameters())

    @param_values.setter
    def param_values(self, param: to.Tensor):
        current = self.param_values
        if param.shape != current.shape:
            raise ShapeErr(given=param, expected_match=current)
        cp.vector_to_parameters(param.to(current.dtype), self.parameters())

    # what subclasses provide ------------------------------------------------------------------------------------------
    @abstractmethod
    def init_param(self, init_values: to.Tensor = None, **kwargs):
        raise NotImplementedError

    @abstractmethod
    def forward(self, obs: to.Tensor) -> to.Tensor:
        raise NotImplementedError

    def reset(self):
        """ Nothing to reset in a feed-forward policy. """

    def evaluate(self, rollout, hidden_states_name: str = 'hidden_states') -> to.Tensor:
        """ The actions for all observations of a rollout but the last (:131-156, feed-forward case). """
        self.eval()
        return self(rollout.get_data_values('observations', truncate_last=True))

    # the fused path ---------------------------------------------------------------------------------------------------
    def fused_spec(self):
        """ Description of this policy for the fused rollout kernels, or None when it has to be evaluated step by
        step through torch. """
        return None

    def fused_params(self) -> to.Tensor:
        """ The parameter vector the fused kernels read (the layout psh_policy.params documents). """
        return self.param_values
