"""
UniformNoise (Pyrado/pyrado/exploration/uniform_noise.py:8-110): learnable additive uniform noise, parametrised by the
log of its half span.  Unlike the normal noise, the floor also applies when the half span is read.
"""
import torch as to
from torch.distributions import Uniform

from ._noise_base import LogScaleNoise


class UniformNoise(LogScaleNoise):
    scale_name = 'halfspan'

    def __init__(self, noise_dim, halfspan_init, halfspan_min=0.01, train_mean: bool = False, learnable: bool = True):
        super().__init__(noise_dim, halfspan_init, halfspan_min, train_mean, learnable, check_init_shape=False)

    @property
    def log_halfspan_init(self) -> to.Tensor:
        return self._log_init

    @property
    def halfspan_min(self) -> to.Tensor:
        return self._floor

    @property
    def halfspan(self) -> to.Tensor:
        return to.max(to.exp(self.log_halfspan), self._floor)

    @halfspan.setter
    def halfspan(self, halfspan: to.Tensor):
        self._set_scale(halfspan)

    def adapt(self, mean: to.Tensor = None, halfspan=None):
        self._adapt(mean, halfspan)

    def forward(self, value: to.Tensor) -> Uniform:
        center, h = self._center(value), self.halfspan
        return Uniform(low=center - h, high=center + h)

    def get_entropy(self) -> to.Tensor:
        return to.log(2*self.halfspan)
