"""
Domain-randomization wrappers (Pyrado/pyrado/environment_wrappers/domain_randomization.py:13-229):
DomainRandWrapperLive draws new domain parameters at every reset, DomainRandWrapperBuffer cycles through a buffer,
MetaDomainRandWrapper adapts the distribution parameters of a wrapped randomizer from a vector.
"""
from typing import Dict, Tuple

import numpy as np

from ..domain_randomization.domain_randomizer import DomainRandomizer
from ..environments.sim_base import SimEnv
from ..utils.exceptions import TypeErr
from .base import EnvWrapper
from .utils import all_envs, inner_env, remove_env


class DomainRandWrapper(EnvWrapper):
    def __init__(self, wrapped_env, randomizer):
        if not isinstance(inner_env(wrapped_env), SimEnv):
            raise TypeErr(given=wrapped_env, expected_type=SimEnv)
        if not isinstance(randomizer, DomainRandomizer) and randomizer is not None:
            raise TypeErr(given=randomizer, expected_type=DomainRandomizer)
        super().__init__(wrapped_env)
        self._randomizer = randomizer

    @property
    def randomizer(self) -> DomainRandomizer:
        return self._randomizer

    @randomizer.setter
    def randomizer(self, randomizer: DomainRandomizer):
        if not isinstance(randomizer, DomainRandomizer):
            raise TypeErr(given=randomizer, expected_type=DomainRandomizer)
        self._randomizer = randomizer


class MetaDomainRandWrapper(DomainRandWrapper):
    def __init__(self, wrapped_rand_env: DomainRandWrapper, mapping: Dict[int, Tuple[str, str]]):
        if not isinstance(wrapped_rand_env, DomainRandWrapper):
            raise TypeErr(given=wrapped_rand_env, expected_type=DomainRandWrapper)
        super().__init__(wrapped_rand_env, None)
        self.mapping = mapping

    @property
    def randomizer(self) -> DomainRandomizer:
        return self._wrapped_env.randomizer

    @randomizer.setter
    def randomizer(self, dr: DomainRandomizer):
        self._wrapped_env.randomizer = dr

    def reset(self, init_state: np.ndarray = None, domain_param: dict = None):
        return self._wrapped_env.reset(init_state, domain_param)

    def adapt_randomizer(self, domain_distr_param_values: np.ndarray):
        for i, value in enumerate(np.asarray(domain_distr_param_values).ravel()):
            dp_name, ddp_name = self.mapping.get(i)
            self._wrapped_env.randomizer.adapt_one_distr_param(dp_name, ddp_name, value)


class DomainRandWrapperLive(DomainRandWrapper):
    def reset(self, init_state: np.ndarray = None, domain_param: dict = None):
        if domain_param is None:
            self._randomizer.randomize(num_samples=1)
            domain_param = self._randomizer.get_params(format='dict', dtype='numpy')
        return self._wrapped_env.reset(init_state, domain_param)


class DomainRandWrapperBuffer(DomainRandWrapper):
    def __init__(self, wrapped_env, randomizer: DomainRandomizer):
        super().__init__(wrapped_env, randomizer)
        self._ring_idx = None
        self._buffer = None

    @property
    def ring_idx(self) -> int:
        return self._ring_idx

    @ring_idx.setter
    def ring_idx(self, idx: int):
        assert isinstance(idx, (int, np.integer)) and idx >= 0
        self._ring_idx = int(idx)

    def fill_buffer(self, num_domains: int):
        assert isinstance(num_domains, int) and num_domains >= 0
        self._randomizer.randomize(num_domains)
        self._buffer = self._randomizer.get_params(-1, format='list', dtype='numpy')
        if isinstance(self._buffer, dict):
            self._buffer = [self._buffer]
        self._ring_idx = 0

    @property
    def buffer(self):
        return self._buffer

    @buffer.setter
    def buffer(self, buffer):
        if not isinstance(buffer, (list, dict)):
            raise TypeErr(given=buffer, expected_type=[list, dict])
        self._buffer = buffer

    def reset(self, init_state: np.ndarray = None, domain_param: dict = None) -> np.ndarray:
        if domain_param is None:
            if isinstance(self._buffer, dict):
                domain_param = self._buffer
            elif isinstance(self._buffer, list):
                domain_param = self._buffer[self._ring_idx]
                self._ring_idx = (self._ring_idx + 1) % len(self._buffer)
            else:
                raise TypeErr(given=self._buffer, given_name='self._buffer', expected_type=[dict, list])
        else:
            self._load_domain_param(domain_param)
        self._wrapped_env.domain_param = domain_param
        return self._wrapped_env.reset(init_state, domain_param=None)


def remove_all_dr_wrappers(env, verbose: bool = False):
    while any(isinstance(e, DomainRandWrapper) for e in all_envs(env)):
        env = remove_env(env, DomainRandWrapper)
    return env
