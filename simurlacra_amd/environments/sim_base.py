"""
SimEnv: simulated environments with domain parameters (Pyrado/pyrado/environments/sim_base.py:10-88).
"""
from abc import abstractmethod

from .base import Env


class SimEnv(Env):
    @property
    @abstractmethod
    def init_space(self):
        raise NotImplementedError

    @classmethod
    @abstractmethod
    def get_nominal_domain_param(cls) -> dict:
        raise NotImplementedError

    @property
    def supported_domain_param(self):
        return self.get_nominal_domain_param().keys()

    @property
    @abstractmethod
    def domain_param(self) -> dict:
        raise NotImplementedError

    @domain_param.setter
    @abstractmethod
    def domain_param(self, param: dict):
        raise NotImplementedError
