"""
DiagNormalNoise (Pyrado/pyrado/exploration/normal_noise.py:10-118): learnable additive Gaussian noise with a diagonal
covariance, parametrised by its log standard deviation.
"""
import math

import torch as to
import torch.nn as nn
from torch.distributions import Normal

from ..utils.exceptions import ShapeErr, TypeErr, ValueErr


class DiagNormalNoise(nn.Module):
    def __init__(self, noise_dim, std_init, std_min=0.01, train_mean: bool = False, learnable: bool = True):
        if not isinstance(std_init, (float, to.Tensor)):
            raise TypeErr(given=std_init, expected_type=(float, to.Tensor))
        if isinstance(std_init, to.Tensor) and not std_init.size() == noise_dim:
            raise ShapeErr(given=std_init, expected_match=to.empty(noise_dim))
        if not (isinstance(std_init, float) and std_init > 0 or isinstance(std_init, to.Tensor) and all(std_init > 0)):
            raise ValueErr(given=std_init, g_constraint='0')
        if not isinstance(std_min, (float, to.Tensor)):
            raise TypeErr(given=std_min, expected_type=(float, to.Tensor))
        super().__init__()
        if learnable:
            self.log_std = nn.Parameter(to.empty(noise_dim, dtype=to.float64), requires_grad=True)
            self.mean = nn.Parameter(to.empty(noise_dim, dtype=to.float64), requires_grad=True) if train_mean else None
        else:
            self.log_std = to.empty(noise_dim, dtype=to.float64)
            self.mean = None
        self.log_std_init = to.log(to.tensor(std_init, dtype=to.float64)) if isinstance(std_init, float) else to.log(std_init.to(to.float64))
        self.std_min = to.tensor(std_min, dtype=to.float64) if isinstance(std_min, float) else std_min.to(to.float64)
        self.reset_expl_params()

    @property
    def std(self) -> to.Tensor:
        return to.exp(self.log_std)

    @std.setter
    def std(self, std: to.Tensor):
        self.log_std.data = to.log(to.max(std, self.std_min))

    def reset_expl_params(self):
        if self.mean is not None:
            self.mean.data.zero_()
        self.log_std.data.copy_(self.log_std_init)

    def adapt(self, mean: to.Tensor = None, std=None):
        if not (isinstance(mean, to.Tensor) or mean is None):
            raise TypeErr(given=mean, expected_type=to.Tensor)
        if not (isinstance(std, to.Tensor) and (std >= 0).all() or std is None):
            raise TypeErr(msg='The std must be a Tensor with all elements > 0 or None!')
        if mean is not None:
            assert self.mean is not None, 'Can not change fixed zero mean!'
            if not mean.shape == self.mean.shape:
                raise ShapeErr(given=mean, expected_match=self.mean)
            self.mean.data = mean
        if std is not None:
            if not std.shape == self.log_std.shape:
                raise ShapeErr(given=std, expected_match=self.std)
            self.std = std

    def forward(self, value: to.Tensor) -> Normal:
        mean = value if self.mean is None else value + self.mean
        return Normal(mean, self.std)

    def get_entropy(self) -> to.Tensor:
        return 0.5 + 0.5*to.log(to.tensor(2*math.pi)) + 0.5*to.log(to.prod(to.pow(self.std, 2)))
