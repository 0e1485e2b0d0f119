"""
GaussianObsNoiseWrapper (Pyrado/pyrado/environment_wrappers/observation_noise.py:8-61): i.i.d. normal noise added to
every observation the wrapped environment returns.
"""
import numpy as np

from .base import EnvWrapperObs


class GaussianObsNoiseWrapper(EnvWrapperObs):
    def __init__(self, wrapped_env, noise_mean=None, noise_std=None):
        super().__init__(wrapped_env)
        if noise_mean is not None:
            self._mean = np.array(noise_mean)
            assert self._mean.shape == self.obs_space.shape
        else:
            self._mean = np.zeros(self.obs_space.shape)
        if noise_std is not None:
            self._std = np.array(noise_std)
            assert self._std.shape == self.obs_space.shape
        else:
            self._std = np.zeros(self.obs_space.shape)

    def _process_obs(self, obs: np.ndarray) -> np.ndarray:
        noise = np.random.randn(*self.obs_space.shape)*self._std + self._mean
        return obs + noise

    def _save_domain_param(self, domain_param: dict):
        domain_param['obs_noise_mean'] = self._mean
        domain_param['obs_noise_std'] = self._std

    def _load_domain_param(self, domain_param: dict):
        if 'obs_noise_mean' in domain_param:
            self._mean = np.array(domain_param['obs_noise_mean'])
            assert self._mean.shape == self.obs_space.shape
        if 'obs_noise_std' in domain_param:
            self._std = np.array(domain_param['obs_noise_std'])
            assert self._std.shape == self.obs_space.shape
