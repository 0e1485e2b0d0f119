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
    name: str = None

    def __init__(self, spec: EnvSpec, use_cuda: bool = False):
        super().__init__()
        if not isinstance(spec, EnvSpec):
            raise TypeErr(given=spec, expected_type=EnvSpec)
        self._env_spec = spec
        self._device = 'cuda' if use_cuda and to.cuda.is_available() else 'cpu'

    @property
    def device(self) -> str:
        return self._device

    @property
    def env_spec(self) -> EnvSpec:
        return self._env_spec

    @property
    def param_values(self) -> to.Tensor:
        return cp.parameters_to_vector(self.parameters())

    @param_values.setter
    def param_values(self, param: to.Tensor):
        if not self.param_values.shape == param.shape:
            raise ShapeErr(given=param, expected_match=self.param_values)
        cp.vector_to_parameters(param.to(self.param_values.dtype), self.parameters())

    @property
    def num_param(self) -> int:
        return sum(p.data.numel() for p in self.parameters())

    @property
    def is_recurrent(self) -> bool:
        return False

    @abstractmethod
    def init_param(self, init_values: to.Tensor = None, **kwargs):
        raise NotImplementedError

    def reset(self):
        pass

    @abstractmethod
    def forward(self, obs: to.Tensor) -> to.Tensor:
        raise NotImplementedError

    def evaluate(self, rollout, hidden_states_name: str = 'hidden_states') -> to.Tensor:
        self.eval()
        return self(rollout.get_data_values('observations', truncate_last=True))

    def fused_spec(self):
        """ Description of this policy for the fused rollout kernels, or None when it has to be evaluated step by
        step through torch. """
        return None

    def fused_params(self) -> to.Tensor:
        """ The parameter vector the fused kernels read (the layout psh_policy.params documents). """
        return self.param_values
