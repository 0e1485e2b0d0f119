from .domain_parameter import (BernoulliDomainParam, DomainParam, MultivariateNormalDomainParam, NormalDomainParam,
                               UniformDomainParam)
from .domain_randomizer import DomainRandomizer
from .default_randomizers import (get_conservative_randomizer, get_default_domain_param_map_omo,
                                  get_default_domain_param_map_qq, get_default_randomizer, get_default_randomizer_bob,
                                  get_default_randomizer_omo, get_default_randomizer_pend, get_default_randomizer_qbb, get_default_randomizer_qcp,
                                  get_default_randomizer_qq, get_empty_randomizer,
                                  get_uniform_masses_lengths_randomizer_qq, get_zero_var_randomizer)
