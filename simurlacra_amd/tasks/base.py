"""
Task and TaskWrapper interface (Pyrado/pyrado/tasks/base.py:12-223).
"""
from abc import ABC, abstractmethod

import numpy as np

from ..utils.exceptions import TypeErr


def never_succeeded(err_state: np.ndarray) -> bool:
    """ Default success function of the desired-state tasks (Pyrado/pyrado/tasks/utils.py:4-6). """
    return False


class Task(ABC):
    @property
    def env_spec(self):
        raise NotImplementedError

    @property
    def state_des(self):
        raise AttributeError(f'{type(self).__name__} has no desired state.')

    @property
    @abstractmethod
    def rew_fcn(self):
        raise NotImplementedError

    def reset(self, **kwargs):
        raise NotImplementedError

    @abstractmethod
    def step_rew(self, state, act, remaining_steps) -> float:
        raise NotImplementedError

    def final_rew(self, state, remaining_steps) -> float:
        return sum(t.compute_final_rew(state, remaining_steps) for t in all_tasks(self))

    def compute_final_rew(self, state, remaining_steps) -> float:
        return 0.

    @abstractmethod
    def has_succeeded(self, state) -> bool:
        raise NotImplementedError

    def has_failed(self, state) -> bool:
        return not self.env_spec.state_space.contains(state)

    def is_done(self, state) -> bool:
        return self.has_succeeded(state) or self.has_failed(state)


def all_tasks(task):
    yield task
    while isinstance(task, TaskWrapper):
        task = task.wrapped_task
        yield task


def _hand_down(name: str):
    """ A method that calls the method of that name on the wrapped task. """
    def method(self, *args, **kwargs):
        return getattr(self._wrapped_task, name)(*args, **kwargs)
    method.__name__ = name
    return method


class TaskWrapper(Task):
    """ Passes everything on to the task it wraps (:139-223); subclasses change what they are there for -- the final
    reward, say -- and inherit the rest. """

    def __init__(self, wrapped_task: Task):
        if not isinstance(wrapped_task, Task):
            raise TypeErr(given=wrapped_task, expected_type=Task)
        self._wrapped_task = wrapped_task

    wrapped_task = property(lambda self: self._wrapped_task)
    env_spec = property(lambda self: self._wrapped_task.env_spec)
    rew_fcn = property(lambda self: self._wrapped_task.rew_fcn)
    state_des = property(lambda self: self._wrapped_task.state_des,
                         lambda self, value: setattr(self._wrapped_task, 'state_des', value))

    reset = _hand_down('reset')
    has_succeeded = _hand_down('has_succeeded')
    has_failed = _hand_down('has_failed')
    is_done = _hand_down('is_done')

    def step_rew(self, state, act, remaining_steps=None):
        return self._wrapped_task.step_rew(state, act, remaining_steps)
