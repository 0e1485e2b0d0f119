"""
Shared part of the two Gaussian noise wrappers: a mean and a standard-deviation vector shaped like the space the noise
lives in, and the draw itself.  The fused rollout kernels read `_mean` / `_std` (EnvPlan.wrap_spec) and draw from their
own counter-based generator; the draw below serves single-environment stepping on the host.
"""
import numpy as np

from ..utils.exceptions import ShapeErr


def noise_vector(value, shape, fill=0.):
    """ `value` as an array of `shape` (None -> constant `fill`); ShapeErr for anything else """
    if value is None:
        return np.full(shape, fill, dtype=np.float64)
    vec = np.array(value, dtype=np.float64)
    if vec.shape != tuple(shape):
        raise ShapeErr(msg=f'The noise parameter has shape {vec.shape}, the space it acts on has shape {tuple(shape)}!')
    return vec


class GaussianNoiseMixin:
    """ Expects `self._noise_shape()`; provides `_mean`, `_std` and `_perturb(x)` = x + N(mean, std^2). """

    def _init_noise(self, noise_mean, noise_std):
        shape = self._noise_shape()
        self._mean, self._std = noise_vector(noise_mean, shape), noise_vector(noise_std, shape)

    def _perturb(self, x: np.ndarray) -> np.ndarray:
        return x + (np.random.standard_normal(self._noise_shape())*self._std + self._mean)
