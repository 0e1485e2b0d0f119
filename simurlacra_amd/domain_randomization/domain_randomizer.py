"""
DomainRandomizer (Pyrado/pyrado/domain_randomization/domain_randomizer.py:13-212).
"""
from copy import deepcopy

from ..utils.exceptions import TypeErr, ValueErr
from .domain_parameter import DomainParam


class DomainRandomizer:
    def __init__(self, *domain_params: DomainParam):
        self.domain_params = []
        self.add_domain_params(*domain_params)
        self._params_pert_dict = None
        self._params_pert_list = None

    def add_domain_params(self, *domain_params: DomainParam):
        for dp in domain_params:
            if not isinstance(dp, DomainParam):
                raise TypeErr(given=dp, expected_type=DomainParam)
            self.domain_params.append(dp)

    def randomize(self, num_samples: int):
        assert isinstance(num_samples, int) and num_samples > 0
        keys = [dp.name for dp in self.domain_params]
        values = [dp.sample(num_samples) for dp in self.domain_params]
        self._params_pert_dict = dict(zip(keys, values))
        self._params_pert_list = [{k: v[i] for k, v in zip(keys, values)} for i in range(num_samples)]

    @staticmethod
    def _conv(d, dtype):
        return {k: (v.numpy() if dtype == 'numpy' else v) for k, v in d.items()}

    def get_params(self, num_samples: int = -1, format: str = 'list', dtype: str = 'numpy'):
        assert isinstance(num_samples, int) and num_samples > -2 and num_samples != 0
        assert format.lower() in ('list', 'dict') and dtype.lower() in ('numpy', 'torch')
        lst = self._params_pert_list
        if num_samples == -1 and len(lst) > 1:
            if format == 'list':
                return [self._conv(deepcopy(d), dtype) for d in lst]
            d = deepcopy(self._params_pert_dict)
            return {k: [s.numpy() if dtype == 'numpy' else s for s in v] for k, v in d.items()}
        if num_samples == 1 or len(lst) == 1:
            return self._conv(deepcopy(lst[0]), dtype)
        if format == 'list':
            return [self._conv(deepcopy(d), dtype) for d in lst[:num_samples]]
        return {k: [s.numpy() if dtype == 'numpy' else s for s in v[:num_samples]]
                for k, v in self._params_pert_dict.items()}

    def adapt_one_distr_param(self, domain_param_name: str, domain_distr_param: str, domain_distr_param_value):
        for dp in self.domain_params:
            if dp.name == domain_param_name:
                if domain_distr_param not in dp.get_field_names():
                    raise KeyError(f'The domain parameter {dp.name} does not have a domain distribution parameter '
                                   f'called {domain_distr_param}!')
                dp.adapt(domain_distr_param, domain_distr_param_value)

    def rescale_distr_param(self, param: str, scale: float):
        if not scale >= 0:
            raise ValueErr(given=scale, ge_constraint='0')
        for dp in self.domain_params:
            if hasattr(dp, param):
                setattr(dp, param, scale*getattr(dp, param))
            dp.distr = dp._make_distr()

    def device_table(self, row_names, extra_rows=()):
        """ Distribution entries for the on-device sampler: one per domain parameter whose name is a row of the
        table (`row_names`, then `extra_rows` such as 'act_delay').  Parameters with other names (the reference's
        default QBB randomizer holds one called ']') are ignored by the simulation and therefore skipped. """
        rows = list(row_names) + list(extra_rows)
        return [dp.to_distr(rows.index(dp.name)) for dp in self.domain_params if dp.name in rows]
