"""
SimEnv: simulated environments with domain parameters (Pyrado/pyrado/environments/sim_base.py:10-88).
"""
from abc import abstractmethod

from .base import Env, required


def _undefined(self, *value):
    raise NotImplementedError(f'{type(self).__name__}.domain_param')


class SimEnv(Env):
    """ An environment whose physics parameters can be read and set as a dict, and whose initial states come from a space. """
    init_space = required('init_space', 'space the initial states are drawn from')
    domain_param = property(abstractmethod(_undefined), abstractmethod(_undefined), doc='the physics parameters as a dict')
    supported_domain_param = property(lambda self: self.get_nominal_domain_param().keys())

    @classmethod
    @abstractmethod
    def get_nominal_domain_param(cls) -> dict:
        raise NotImplementedError
