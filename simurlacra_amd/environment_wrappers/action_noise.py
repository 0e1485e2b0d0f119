"""
GaussianActNoiseWrapper (Pyrado/pyrado/environment_wrappers/action_noise.py:9-71): i.i.d. normal noise added to every
action before it reaches the wrapped environment.
"""
import numpy as np

from ..utils.exceptions import ShapeErr
from .base import EnvWrapperAct


class GaussianActNoiseWrapper(EnvWrapperAct):
    def __init__(self, wrapped_env, noise_mean=None, noise_std=None):
        super().__init__(wrapped_env)
        if noise_mean is not None:
            self._mean = np.array(noise_mean)
            if not self._mean.shape == self.act_space.shape:
                raise ShapeErr(given=self._mean, expected_match=self.act_space)
        else:
            self._mean = np.zeros(self.act_space.shape)
        if noise_std is not None:
            self._std = np.array(noise_std)
            if not self._std.shape == self.act_space.shape:
                raise ShapeErr(given=self._std, expected_match=self.act_space)
        else:
            self._std = np.zeros(self.act_space.shape)

    def _process_act(self, act: np.ndarray) -> np.ndarray:
        noise = np.random.randn(*self.act_space.shape)*self._std + self._mean
        return act + noise

    def _save_domain_param(self, domain_param: dict):
        domain_param['act_noise_mean'] = self._mean
        domain_param['act_noise_std'] = self._std

    def _load_domain_param(self, domain_param: dict):
        # As in the reference (:58-71) the loaded values land in attributes that _process_act does not read, so
        # noise parameters passed through the domain parameters have no effect.
        if 'act_noise_mean' in domain_param:
            self._noise_mean = np.array(domain_param['act_noise_mean'])
            if not self._noise_mean.shape == self.act_space.shape:
                raise ShapeErr(given=self._noise_mean, expected_match=self.act_space)
        if 'act_noise_std' in domain_param:
            self._noise_std = np.array(domain_param['act_noise_std'])
            if not self._noise_std.shape == self.act_space.shape:
                raise ShapeErr(given=self._noise_std, expected_match=self.act_space)
