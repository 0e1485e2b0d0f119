from .base import EnvWrapper, EnvWrapperAct, EnvWrapperObs
from .action_delay import ActDelayWrapper
from .action_noise import GaussianActNoiseWrapper
from .action_normalization import ActNormWrapper
from .downsampling import DownsamplingWrapper
from .domain_randomization import (DomainRandWrapper, DomainRandWrapperBuffer, DomainRandWrapperLive,
                                   MetaDomainRandWrapper, remove_all_dr_wrappers)
from .observation_noise import GaussianObsNoiseWrapper
from .observation_normalization import ObsNormWrapper, ObsRunningNormWrapper
from .observation_partial import ObsPartialWrapper
from .utils import all_envs, attr_env, attr_env_get, inner_env, remove_env, typed_env
