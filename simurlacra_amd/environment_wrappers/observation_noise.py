"""
GaussianObsNoiseWrapper (Pyrado/pyrado/environment_wrappers/observation_noise.py:8-61): i.i.d. normal noise added to
every observation the wrapped environment returns; mean and standard deviation are domain parameters.
"""
import numpy as np

from ._noise import GaussianNoiseMixin, noise_vector
from .base import EnvWrapperObs


class GaussianObsNoiseWrapper(GaussianNoiseMixin, EnvWrapperObs):
    def __init__(self, wrapped_env, noise_mean=None, noise_std=None):
        EnvWrapperObs.__init__(self, wrapped_env)
        self._init_noise(noise_mean, noise_std)

    def _noise_shape(self):
        return self.obs_space.shape

    def _process_obs(self, obs: np.ndarray) -> np.ndarray:
        return self._perturb(obs)

    def _save_domain_param(self, domain_param: dict):
        domain_param.update(obs_noise_mean=self._mean, obs_noise_std=self._std)

    def _load_domain_param(self, domain_param: dict):
        if 'obs_noise_mean' in domain_param:
            self._mean = noise_vector(domain_param['obs_noise_mean'], self._noise_shape())
        if 'obs_noise_std' in domain_param:
            self._std = noise_vector(domain_param['obs_noise_std'], self._noise_shape())
