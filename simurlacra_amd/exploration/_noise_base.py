"""
Common machinery of the additive exploration-noise modules: a positive, per-dimension scale kept as its logarithm
(learnable or fixed), floored at a minimum when set, and an optional learnable mean.  Subclasses name the scale
(`std`, `halfspan`), build the distribution around a value and state its entropy.
"""
import torch as to
import torch.nn as nn

from ..utils.exceptions import ShapeErr, TypeErr, ValueErr


def _positive(value, name, noise_dim=None):
    """ float or tensor > 0 as an fp64 tensor; the reference's error types for anything else """
    if not isinstance(value, (float, to.Tensor)):
        raise TypeErr(given=value, expected_type=(float, to.Tensor))
    t = to.as_tensor(value, dtype=to.float64)
    if noise_dim is not None and isinstance(value, to.Tensor) and value.size() != to.Size(noise_dim if isinstance(noise_dim, (tuple, list, to.Size)) else (noise_dim,)):
        raise ShapeErr(given=value, expected_match=to.empty(noise_dim))
    if not bool((t > 0).all()):
        raise ValueErr(given=value, given_name=name, g_constraint='0')
    return t


class LogScaleNoise(nn.Module):
    scale_name = 'scale'          # attribute name of the log-scale tensor is 'log_' + scale_name

    def __init__(self, noise_dim, scale_init, scale_min, train_mean: bool, learnable: bool, check_init_shape: bool):
        init = _positive(scale_init, self.scale_name + '_init', noise_dim if check_init_shape else None)
        floor = _positive(scale_min, self.scale_name + '_min')
        super().__init__()
        log_scale = to.empty(noise_dim, dtype=to.float64)
        if learnable:
            log_scale = nn.Parameter(log_scale, requires_grad=True)
        setattr(self, 'log_' + self.scale_name, log_scale)
        self.mean = nn.Parameter(to.empty(noise_dim, dtype=to.float64), requires_grad=True) if learnable and train_mean else None
        self._log_init, self._floor = to.log(init), floor
        self.reset_expl_params()

    @property
    def _log_scale(self) -> to.Tensor:
        return getattr(self, 'log_' + self.scale_name)

    def _set_scale(self, scale: to.Tensor):
        self._log_scale.data = to.log(to.max(scale, self._floor))

    def reset_expl_params(self):
        if self.mean is not None:
            self.mean.data.zero_()
        self._log_scale.data.copy_(self._log_init)

    def _adapt(self, mean, scale):
        if not (mean is None or isinstance(mean, to.Tensor)):
            raise TypeErr(given=mean, expected_type=to.Tensor)
        if not (scale is None or (isinstance(scale, to.Tensor) and bool((scale >= 0).all()))):
            raise TypeErr(msg=f'The {self.scale_name} must be a Tensor with all elements > 0 or None!')
        if mean is not None:
            assert self.mean is not None, 'Can not change fixed zero mean!'
            if mean.shape != self.mean.shape:
                raise ShapeErr(given=mean, expected_match=self.mean)
            self.mean.data = mean
        if scale is not None:
            if scale.shape != self._log_scale.shape:
                raise ShapeErr(given=scale, expected_match=self._log_scale)
            self._set_scale(scale)

    def _center(self, value: to.Tensor) -> to.Tensor:
        return value if self.mean is None else value + self.mean
