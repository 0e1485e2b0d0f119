"""
DiagNormalNoise (Pyrado/pyrado/exploration/normal_noise.py:10-118): learnable additive Gaussian noise with a diagonal
covariance, parametrised by its log standard deviation.
"""
import math

import torch as to
from torch.distributions import Normal

from ._noise_base import LogScaleNoise


class DiagNormalNoise(LogScaleNoise):
    scale_name = 'std'

    def __init__(self, noise_dim, std_init, std_min=0.01, train_mean: bool = False, learnable: bool = True):
        super().__init__(noise_dim, std_init, std_min, train_mean, learnable, check_init_shape=True)

    @property
    def log_std_init(self) -> to.Tensor:
        return self._log_init

    @property
    def std_min(self) -> to.Tensor:
        return self._floor

    @property
    def std(self) -> to.Tensor:
        return to.exp(self.log_std)

    @std.setter
    def std(self, std: to.Tensor):
        self._set_scale(std)

    def adapt(self, mean: to.Tensor = None, std=None):
        self._adapt(mean, std)

    def forward(self, value: to.Tensor) -> Normal:
        return Normal(self._center(value), self.std)

    def get_entropy(self) -> to.Tensor:
        # differential entropy of N(., diag(std^2)) written as in the reference: 1/2 + ln(2 pi)/2 + ln(prod std^2)/2
        return 0.5 + 0.5*math.log(2*math.pi) + 0.5*to.log(to.prod(self.std**2))
