"""
FinalRewTask and FinalRewMode (Pyrado/pyrado/tasks/final_reward.py:12-198): a reward given once when the episode ends.
"""
from typing import NamedTuple

import numpy as np

from ..utils.exceptions import TypeErr
from .base import Task, TaskWrapper


class FinalRewMode(NamedTuple):
    state_dependent: bool = False
    time_dependent: bool = False
    always_positive: bool = False
    always_negative: bool = False
    user_input: bool = False


class FinalRewTask(TaskWrapper):
    def __init__(self, wrapped_task: Task, mode: FinalRewMode, factor: float = 1e3):
        super().__init__(wrapped_task)
        if not isinstance(mode, FinalRewMode):
            raise TypeErr(given=mode, expected_type=FinalRewMode)
        self.mode, self.factor = mode, factor
        self._yielded_final_rew = False

    @property
    def yielded_final_rew(self) -> bool:
        return self._yielded_final_rew

    def reset(self, **kwargs):
        super().reset(**kwargs)
        self._yielded_final_rew = False

    def compute_final_rew(self, state: np.ndarray, remaining_steps: int) -> float:
        if self._yielded_final_rew:
            return 0.
        self._yielded_final_rew = True
        m = self.mode
        if m.user_input:
            raise NotImplementedError('interactive final rewards are not supported')
        key = (m.always_positive, m.always_negative, m.state_dependent, m.time_dependent)
        failed, succeeded = self.has_failed(state), False
        if not failed:
            succeeded = self.has_succeeded(state)
        if not failed and not succeeded:
            return 0.

        def scale():
            act = np.zeros(self.env_spec.act_space.shape)
            return abs(self._wrapped_task.step_rew(state, act, remaining_steps))

        # (always_positive, always_negative, state_dependent, time_dependent) -> (on failure, on success)
        table = {
            (False, False, False, False): (lambda: -abs(self.factor), lambda: abs(self.factor)),
            (True, False, False, False): (lambda: 0., lambda: abs(self.factor)),
            (False, True, False, False): (lambda: -abs(self.factor), lambda: 0.),
            (True, False, True, False): (lambda: 0., lambda: self.factor*scale()),
            (False, True, True, False): (lambda: -self.factor*scale(), lambda: 0.),
            (False, False, True, False): (lambda: -self.factor*scale(), lambda: self.factor*scale()),
            (False, False, True, True): (lambda: -remaining_steps*scale(), lambda: remaining_steps*scale()),
            (False, False, False, True): (lambda: -1.*remaining_steps, lambda: 1.*remaining_steps),
            (True, False, False, True): (lambda: 0., lambda: 1.*remaining_steps),
            (False, True, False, True): (lambda: -1.*remaining_steps, lambda: 0.),
        }
        if key not in table:
            raise NotImplementedError(f'No matching configuration found for the given FinalRewMode: {self.mode}')
        on_fail, on_succ = table[key]
        return on_fail() if failed else on_succ()
