"""
TimePolicy (Pyrado/pyrado/policies/time.py:11-46): an open-loop policy, action = f(time since reset).

It carries a clock, so it is not one of the policies the fused kernels evaluate: rollout() runs it through the
step-by-step loop (env.step() is then one psh_step launch per step).
"""
from typing import Callable, Sequence

import torch as to

from .base import Policy


class TimePolicy(Policy):
    name: str = 'time'

    def __init__(self, spec, fcn_of_time: Callable[[float], Sequence[float]], dt: float, use_cuda: bool = False):
        super().__init__(spec, use_cuda)
        self._fcn_of_time = fcn_of_time
        self._dt = dt
        self._curr_time = None   # set by reset(), as in the reference: calling the policy before a reset is an error

    def init_param(self, init_values: to.Tensor = None, **kwargs):
        pass

    def reset(self):
        self._curr_time = 0

    def forward(self, obs: to.Tensor = None) -> to.Tensor:
        # the clock accumulates dt in floating point (time.py:45), so f sees 0, dt, dt+dt, ... and not k*dt
        now, self._curr_time = self._curr_time, self._curr_time + self._dt
        return to.as_tensor(self._fcn_of_time(now), dtype=to.float64)
