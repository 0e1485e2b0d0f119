"""
UniformNoise (Pyrado/pyrado/exploration/uniform_noise.py:8-110): learnable additive uniform noise, parametrised by the
log of its half span.
"""
import torch as to
import torch.nn as nn
from torch.distributions import Uniform

from ..utils.exceptions import ShapeErr, TypeErr, ValueErr


class UniformNoise(nn.Module):
    def __init__(self, noise_dim, halfspan_init, halfspan_min=0.01, train_mean: bool = False, learnable: bool = True):
        if not isinstance(halfspan_init, (float, to.Tensor)):
            raise TypeErr(given=halfspan_init, expected_type=[float, to.Tensor])
        if not (isinstance(halfspan_init, float) and halfspan_init > 0 or
                isinstance(halfspan_init, to.Tensor) and all(halfspan_init > 0)):
            raise ValueErr(given=halfspan_init, g_constraint='0')
        if not isinstance(halfspan_min, (float, to.Tensor)):
            raise TypeErr(given=halfspan_min, expected_type=[float, to.Tensor])
        if not (isinstance(halfspan_min, float) and halfspan_min > 0 or
                isinstance(halfspan_min, to.Tensor) and all(halfspan_min > 0)):
            raise ValueErr(given=halfspan_min, g_constraint='0')
        super().__init__()
        if learnable:
            self.log_halfspan = nn.Parameter(to.empty(noise_dim, dtype=to.float64), requires_grad=True)
            self.mean = nn.Parameter(to.empty(noise_dim, dtype=to.float64), requires_grad=True) if train_mean else None
        else:
            self.log_halfspan = to.empty(noise_dim, dtype=to.float64)
            self.mean = None
        self.log_halfspan_init = (to.log(to.tensor(halfspan_init, dtype=to.float64)) if isinstance(halfspan_init, float)
                                  else to.log(halfspan_init.to(to.float64)))
        self.halfspan_min = (to.tensor(halfspan_min, dtype=to.float64) if isinstance(halfspan_min, float)
                             else halfspan_min.to(to.float64))
        self.reset_expl_params()

    @property
    def halfspan(self) -> to.Tensor:
        return to.max(to.exp(self.log_halfspan), self.halfspan_min)

    @halfspan.setter
    def halfspan(self, halfspan: to.Tensor):
        self.log_halfspan.data = to.log(to.max(halfspan, self.halfspan_min))

    def reset_expl_params(self):
        if self.mean is not None:
            self.mean.data.zero_()
        self.log_halfspan.data.copy_(self.log_halfspan_init)

    def adapt(self, mean: to.Tensor = None, halfspan=None):
        if not (isinstance(mean, to.Tensor) or mean is None):
            raise TypeErr(given=mean, expected_type=to.Tensor)
        if not (isinstance(halfspan, to.Tensor) and (halfspan >= 0).all() or halfspan is None):
            raise TypeErr(msg='The halfspan must be a Tensor with all elements > 0 or None!')
        if mean is not None:
            assert self.mean is not None, 'Can not change fixed zero mean!'
            if not mean.shape == self.mean.shape:
                raise ShapeErr(given=mean, expected_match=self.mean)
            self.mean.data = mean
        if halfspan is not None:
            if not halfspan.shape == self.log_halfspan.shape:
                raise ShapeErr(given=halfspan, expected_match=self.halfspan)
            self.halfspan = halfspan

    def forward(self, value: to.Tensor) -> Uniform:
        mean = value if self.mean is None else value + self.mean
        return Uniform(low=mean - self.halfspan, high=mean + self.halfspan)

    def get_entropy(self) -> to.Tensor:
        return to.log(2*self.halfspan)
