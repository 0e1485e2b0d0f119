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


def _require(obj, cls, allow_none: bool = False):
    """ The reference's type checks in one place: anything but an instance of cls (or None where allowed) is a TypeErr. """
    if not (isinstance(obj, cls) or (allow_none and obj is None)):
        raise TypeErr(given=obj, expected_type=cls)
    return obj


class DomainRandWrapper(EnvWrapper):
    """ Base of the randomizing wrappers (:13-48): holds the randomizer, wraps a simulation only. """

    def __init__(self, wrapped_env, randomizer):
        _require(inner_env(wrapped_env), SimEnv)
        self._randomizer = _require(randomizer, DomainRandomizer, allow_none=True)
        super().__init__(wrapped_env)

    @property
    def randomizer(self) -> DomainRandomizer:
        return self._randomizer

    @randomizer.setter
    def randomizer(self, randomizer: DomainRandomizer):
        self._randomizer = _require(randomizer, DomainRandomizer)


class MetaDomainRandWrapper(DomainRandWrapper):
    """ Entry i of a vector sets distribution parameter mapping[i] = (domain parameter, distribution parameter) of the
    randomizer of the wrapper below (:51-103); the randomizer itself stays with that wrapper. """

    def __init__(self, wrapped_rand_env: DomainRandWrapper, mapping: Dict[int, Tuple[str, str]]):
        super().__init__(_require(wrapped_rand_env, DomainRandWrapper), None)
        self.mapping = mapping

    @property
    def randomizer(self) -> DomainRandomizer:
        return self._wrapped_env.randomizer

    @randomizer.setter
    def randomizer(self, dr: DomainRandomizer):
        self._wrapped_env.randomizer = dr

    def adapt_randomizer(self, domain_distr_param_values: np.ndarray):
        target = self._wrapped_env.randomizer
        for idx, val in enumerate(np.asarray(domain_distr_param_values).ravel()):
            target.adapt_one_distr_param(*self.mapping.get(idx), val)

    def reset(self, init_state: np.ndarray = None, domain_param: dict = None):
        return self._wrapped_env.reset(init_state, domain_param)


class DomainRandWrapperLive(DomainRandWrapper):
    """ A fresh draw at every reset that is not given parameters (:106-127). """

    def reset(self, init_state: np.ndarray = None, domain_param: dict = None):
        if domain_param is not None:
            return self._wrapped_env.reset(init_state, domain_param)
        self._randomizer.randomize(num_samples=1)
        return self._wrapped_env.reset(init_state, self._randomizer.get_params(format='dict', dtype='numpy'))


class DomainRandWrapperBuffer(DomainRandWrapper):
    """ Resets walk round a buffer of parameter sets, filled once (:130-199).  The buffer is a list of dicts, or one dict
    that every reset uses. """

    def __init__(self, wrapped_env, randomizer: DomainRandomizer):
        super().__init__(wrapped_env, randomizer)
        self._buffer, self._ring_idx = None, None

    # ring position ------------------------------------------------------------------------------------------------
    def _get_ring_idx(self) -> int:
        return self._ring_idx

    def _set_ring_idx(self, idx: int):
        assert isinstance(idx, (int, np.integer)) and idx >= 0
        self._ring_idx = int(idx)

    ring_idx = property(_get_ring_idx, _set_ring_idx)

    # buffer ---------------------------------------------------------------------------------------------------------
    def _get_buffer(self):
        return self._buffer

    def _set_buffer(self, buffer):
        self._buffer = _require(buffer, (list, dict))

    buffer = property(_get_buffer, _set_buffer)

    def fill_buffer(self, num_domains: int):
        assert isinstance(num_domains, int) and num_domains >= 0
        self._randomizer.randomize(num_domains)
        sets = self._randomizer.get_params(-1, format='list', dtype='numpy')
        self._buffer = [sets] if isinstance(sets, dict) else sets   # (a single draw comes back as one dict)
        self._ring_idx = 0

    def _next_buffered(self) -> dict:
        buf = _require(self._buffer, (dict, list))
        if isinstance(buf, dict):
            return buf
        taken, self._ring_idx = buf[self._ring_idx], (self._ring_idx + 1) % len(buf)
        return taken

    def reset(self, init_state: np.ndarray = None, domain_param: dict = None) -> np.ndarray:
        if domain_param is None:
            domain_param = self._next_buffered()
        else:
            self._load_domain_param(domain_param)
        # the parameters go in through the setter, the inner reset then gets none (:193-199)
        self._wrapped_env.domain_param = domain_param
        return self._wrapped_env.reset(init_state, domain_param=None)


def remove_all_dr_wrappers(env, verbose: bool = False):
    """ Strips every randomizing wrapper from the chain (:202-229). """
    while any(isinstance(e, DomainRandWrapper) for e in all_envs(env)):
        env = remove_env(env, DomainRandWrapper)
    return env
