"""
Domain parameters and their distributions (Pyrado/pyrado/domain_randomization/domain_parameter.py:12-236).
`sample()` draws on the host with torch.distributions as the reference does (single-environment API); `to_distr()`
exports the same distribution for the on-device sampler of the batched path (psh_randomize).
"""
from abc import ABC
from typing import Sequence

import torch as to
from torch.distributions.bernoulli import Bernoulli
from torch.distributions.multivariate_normal import MultivariateNormal
from torch.distributions.normal import Normal
from torch.distributions.uniform import Uniform

from .. import _abi as abi, inf
from ..utils.exceptions import TypeErr


def _f64(x):
    return x.to(to.float64) if isinstance(x, to.Tensor) else to.tensor(float(x), dtype=to.float64)


class DomainParam(ABC):
    def __init__(self, name: str, mean, clip_lo=-inf, clip_up=inf, roundint=False):
        self.name, self.mean = name, mean
        self.clip_lo, self.clip_up, self.roundint = clip_lo, clip_up, roundint
        self.distr = None

    def __eq__(self, other):
        if not isinstance(other, DomainParam):
            raise TypeErr(given=other, expected_type=DomainParam)
        return all(getattr(self, fn) == getattr(other, fn) for fn in self.get_field_names())

    @staticmethod
    def get_field_names() -> Sequence[str]:
        raise NotImplementedError

    def _make_distr(self):
        return None

    def adapt(self, domain_distr_param: str, domain_distr_param_value):
        if domain_distr_param not in self.get_field_names():
            raise KeyError(f'The domain parameter {self.name} does not have a domain distribution parameter '
                           f'called {domain_distr_param}!')
        setattr(self, domain_distr_param, domain_distr_param_value)
        self.distr = self._make_distr()

    def _postprocess(self, sample_tensor):
        sample_tensor = to.clamp(sample_tensor, self.clip_lo, self.clip_up)
        if self.roundint:
            sample_tensor = to.round(sample_tensor).type(to.int)
        return list(sample_tensor)

    def sample(self, num_samples: int = 1) -> list:
        assert isinstance(num_samples, int) and num_samples > 0
        if self.distr is None:
            return list(to.ones(num_samples, dtype=to.float64)*self.mean)
        return self._postprocess(self.distr.sample(sample_shape=to.Size([num_samples])))

    def to_distr(self, row: int) -> dict:
        """ Entry of the device randomizer table for table row `row`. """
        raise NotImplementedError(f'{type(self).__name__} has no on-device sampler')

    def _common(self, row):
        return dict(row=row, clip_lo=float(self.clip_lo), clip_up=float(self.clip_up), roundint=int(bool(self.roundint)))


class UniformDomainParam(DomainParam):
    def __init__(self, halfspan: float, **kwargs):
        super().__init__(**kwargs)
        self.halfspan = halfspan
        self.distr = self._make_distr()

    def _make_distr(self):
        if float(self.halfspan) == 0.:
            return None  # a zero-width interval is the constant `mean` (torch rejects Uniform(a, a))
        # no argument validation: the reference's default QBB randomizer holds negative half spans (low > high),
        # which torch samples as low + u*(high - low), i.e. uniformly on the same interval
        return Uniform(_f64(self.mean) - _f64(self.halfspan), _f64(self.mean) + _f64(self.halfspan),
                       validate_args=False)

    @staticmethod
    def get_field_names():
        return ['name', 'mean', 'halfspan', 'clip_lo', 'clip_up', 'roundint']

    def sample(self, num_samples: int = 1) -> list:
        if self.distr is None:
            return self._postprocess(to.ones(num_samples, dtype=to.float64)*float(self.mean))
        return super().sample(num_samples)

    def to_distr(self, row):
        return dict(kind=abi.DISTR_UNIFORM, a=float(self.mean), b=float(self.halfspan), **self._common(row))


class NormalDomainParam(DomainParam):
    def __init__(self, std, **kwargs):
        super().__init__(**kwargs)
        self.std = std
        self.distr = self._make_distr()

    def _make_distr(self):
        if float(self.std) == 0.:
            return None
        return Normal(_f64(self.mean), _f64(self.std))

    @staticmethod
    def get_field_names():
        return ['name', 'mean', 'std', 'clip_lo', 'clip_up', 'roundint']

    def sample(self, num_samples: int = 1) -> list:
        if self.distr is None:
            return self._postprocess(to.ones(num_samples, dtype=to.float64)*float(self.mean))
        return super().sample(num_samples)

    def to_distr(self, row):
        return dict(kind=abi.DISTR_NORMAL, a=float(self.mean), b=float(self.std), **self._common(row))


class MultivariateNormalDomainParam(DomainParam):
    def __init__(self, cov, **kwargs):
        assert len(cov.shape) == 2, 'Covariance needs to be given as a matrix'
        super().__init__(**kwargs)
        self.mean = self.mean.view(-1)
        self.cov = cov
        self.distr = self._make_distr()

    def _make_distr(self):
        return MultivariateNormal(self.mean, self.cov)

    @staticmethod
    def get_field_names():
        return ['name', 'mean', 'cov', 'clip_lo', 'clip_up', 'roundint']


class BernoulliDomainParam(DomainParam):
    def __init__(self, val_0, val_1, prob_1: float, **kwargs):
        kwargs.setdefault('mean', None)
        super().__init__(**kwargs)
        self.val_0, self.val_1, self.prob_1 = val_0, val_1, prob_1
        self.distr = self._make_distr()

    def _make_distr(self):
        return Bernoulli(_f64(self.prob_1))

    @staticmethod
    def get_field_names():
        return ['name', 'mean', 'val_0', 'val_1', 'prob_1', 'clip_lo', 'clip_up', 'roundint']

    def sample(self, num_samples: int = 1) -> list:
        assert isinstance(num_samples, int) and num_samples > 0
        s = self.distr.sample(sample_shape=to.Size([num_samples]))
        return self._postprocess((to.ones_like(s) - s)*self.val_0 + s*self.val_1)

    def to_distr(self, row):
        return dict(kind=abi.DISTR_BERNOULLI, a=float(self.val_0), b=float(self.val_1), c=float(self.prob_1),
                    **self._common(row))
