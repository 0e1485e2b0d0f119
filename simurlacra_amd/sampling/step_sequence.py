"""
StepSequence: the container rollouts are returned in (Pyrado/pyrado/sampling/step_sequence.py:186-695).
Own implementation of the same interface: per-variable arrays stacked along the step axis (observations may hold one
more entry than there are steps), dict-valued variables (env_infos), a `done` array, rollout bounds for concatenated
sequences, slicing, step proxies, numpy/torch conversion and (un)discounted returns.
"""
from collections.abc import Iterable, Sequence
from copy import deepcopy

import numpy as np
import torch as to

from ..utils.exceptions import ShapeErr, ValueErr


def _stack(values, fmt):
    if isinstance(values, (np.ndarray, to.Tensor)):
        return _convert(values, fmt)
    if isinstance(values, dict):  # already a dict of stacked arrays
        return {k: _stack(v, fmt) for k, v in values.items()}
    first = values[0]
    if isinstance(first, dict):
        return {k: _stack([v[k] for v in values], fmt) for k in first}
    if isinstance(first, to.Tensor):
        return _convert(to.stack(list(values)), fmt)
    return _convert(np.stack([np.asarray(v) for v in values]), fmt)


def _convert(value, fmt, dtype=None):
    if isinstance(value, dict):
        return {k: _convert(v, fmt, dtype) for k, v in value.items()}
    if fmt == 'torch':
        t = value if isinstance(value, to.Tensor) else to.from_numpy(np.ascontiguousarray(value))
        return t if dtype is None else t.to(dtype)
    a = value.detach().cpu().numpy() if isinstance(value, to.Tensor) else np.asarray(value)
    return a if dtype is None else a.astype(dtype)


def _length(value):
    if isinstance(value, dict):
        return _length(next(iter(value.values())))
    return value.shape[0] if isinstance(value, (np.ndarray, to.Tensor)) else len(value)


def _index(value, idx):
    if isinstance(value, dict):
        return {k: _index(v, idx) for k, v in value.items()}
    return value[idx]


class Step:
    """ Proxy of one step: `step.observation`, `step.action`, `step.reward`, `step.next_observation`, `step.done`. """

    def __init__(self, seq, index: int):
        self._seq, self._index = seq, index

    def __getattr__(self, name):
        seq, i = self.__dict__['_seq'], self.__dict__['_index']
        nxt = name.startswith('next_')
        base = name[5:] if nxt else name
        for cand in (base, base + 's'):
            if cand in seq.data_names:
                return _index(getattr(seq, cand), i + 1 if nxt else i)
        if base == 'done':
            return seq.done[i]
        raise AttributeError(name)


class StepSequence(Sequence):
    required_fields = {'observations', 'actions'}

    def __init__(self, *, complete: bool = True, rollout_info=None, data_format: str = None, done=None,
                 continuous: bool = True, rollout_bounds=None, rewards, **data):
        self.rollout_info, self.continuous = rollout_info, continuous
        self.length = _length(rewards)
        if self.length == 0:
            raise ValueError('StepSequence cannot be empty!')
        if data_format is None:
            data_format = 'numpy'
            for v in data.values():
                if isinstance(v, to.Tensor) or (isinstance(v, list) and len(v) > 0 and isinstance(v[0], to.Tensor)):
                    data_format = 'torch'
                    break
        self._data_format = data_format
        missing = StepSequence.required_fields - data.keys()
        if missing:
            raise ValueError(f'Missing required data fields: {missing}')
        self._data_names = []
        self.add_data('rewards', rewards)
        for name, value in data.items():
            self.add_data(name, value)
        if done is None:
            done = np.zeros(self.length, dtype=bool)
            if complete and continuous:
                done[-1] = True
        else:
            done = np.asarray(done, dtype=bool)
            assert done.shape[0] == self.length
        self.done = done
        if continuous:
            if rollout_bounds is None:
                rollout_bounds = [0] + list(np.flatnonzero(done) + 1)
                if not done[-1]:
                    rollout_bounds.append(self.length)
            else:
                assert rollout_bounds[0] == 0 and rollout_bounds[-1] == self.length
                assert all(a < b for a, b in zip(rollout_bounds[:-1], rollout_bounds[1:]))
            self._rollout_bounds = np.array(rollout_bounds)
        else:
            self._rollout_bounds = None

    # ------------------------------------------------------------------------------------------ data fields
    @property
    def data_format(self) -> str:
        return self._data_format

    @property
    def data_names(self):
        return self._data_names

    @property
    def rollout_bounds(self):
        return self._rollout_bounds

    def add_data(self, name: str, value=None, item_shape: tuple = None, with_after_last: bool = False):
        if name in self._data_names:
            raise ValueErr(msg=f'Trying to add a duplicate data field for {name}')
        if value is None:
            n = self.length + (1 if with_after_last else 0)
            value = (to.zeros((n,) + tuple(item_shape)) if self._data_format == 'torch'
                     else np.zeros((n,) + tuple(item_shape)))
        else:
            value = _stack(value, self._data_format)
            n = _length(value)
            if isinstance(value, to.Tensor):
                assert not value.requires_grad, 'Do not add gradient-tracking tensors to a StepSequence'
            ok = n == self.length or (self.continuous and n == self.length + 1)
            if not ok:
                raise ShapeErr(msg=f'The data list {name} must have {self.length} or {self.length}+1 elements, '
                                   f'but has {n} elements.')
        self._data_names.append(name)
        self.__dict__[name] = value

    def get_data_values(self, name: str, truncate_last: bool = False):
        assert name in self._data_names
        entry = self.__dict__[name]
        return self._truncate(entry) if truncate_last else entry

    def _truncate(self, entry):
        if isinstance(entry, dict):
            return {k: self._truncate(v) for k, v in entry.items()}
        return entry[:-1] if _length(entry) == self.length + 1 else entry

    # --------------------------------------------------------------------------------------------- sequence
    def __len__(self):
        return self.length

    def __getitem__(self, index):
        if isinstance(index, (slice, Iterable)):
            cont = isinstance(index, slice) and index.step in (None, 1)
            data = {}
            for name in self._data_names:
                entry = self.__dict__[name]
                if cont and _length(entry) == self.length + 1:
                    start, end, _ = index.indices(self.length)
                    data[name] = _index(entry, slice(start, end + 1))
                else:
                    data[name] = _index(self._truncate(entry), index)
            bounds = None
            if cont:
                start, end, _ = index.indices(self.length)
                bounds = [0] + [int(b - start) for b in self._rollout_bounds if start < b < end] + [end - start]
            return StepSequence(rollout_info=self.rollout_info, data_format=self._data_format, done=self.done[index],
                                continuous=cont, rollout_bounds=bounds, **data)
        i = int(index)
        if i < 0:
            i += self.length
        if not 0 <= i < self.length:
            raise IndexError(index)
        return Step(self, i)

    def __iter__(self):
        return (Step(self, i) for i in range(self.length))

    # ------------------------------------------------------------------------------------------- conversion
    def numpy(self, data_type=None):
        self.convert('numpy', data_type)

    def torch(self, data_type=None):
        self.convert('torch', data_type)

    def convert(self, data_format: str, data_type=None):
        if data_format not in {'torch', 'numpy'}:
            raise ValueErr(given=data_format, eq_constraint="'torch' or 'numpy'")
        if self._data_format == data_format and data_type is None:
            return
        self._data_format = data_format
        for name in self._data_names:
            self.__dict__[name] = _convert(self.__dict__[name], data_format, data_type)

    # --------------------------------------------------------------------------------------------- rollouts
    @property
    def rollout_count(self) -> int:
        if not self.continuous:
            raise ValueErr(msg='Sub-rollouts are only supported on continuous data.')
        return len(self._rollout_bounds) - 1

    @property
    def rollout_lengths(self):
        if not self.continuous:
            raise ValueErr(msg='Sub-rollouts are only supported on continuous data.')
        return self._rollout_bounds[1:] - self._rollout_bounds[:-1]

    def get_rollout(self, index):
        if not self.continuous:
            raise ValueErr(msg='Sub-rollouts are only supported on continuous data.')
        if isinstance(index, slice):
            start, end, step = index.indices(self.rollout_count)
            if step == 1:
                return self[int(self._rollout_bounds[start]):int(self._rollout_bounds[end])]
            index = range(start, end, step)
        if isinstance(index, Iterable):
            return StepSequence.concat([self.get_rollout(i) for i in index], self._data_format)
        i = int(index)
        if i < 0:
            i += self.rollout_count
        return self[int(self._rollout_bounds[i]):int(self._rollout_bounds[i + 1])]

    def iterate_rollouts(self):
        if self.rollout_count == 1:
            yield self
        else:
            for i in range(self.rollout_count):
                yield self.get_rollout(i)

    def undiscounted_return(self) -> float:
        if not len(self._rollout_bounds) == 2:
            raise ShapeErr(msg='The StepSequence must be a single continuous rollout.')
        return self.rewards.sum()

    def discounted_return(self, gamma: float):
        if not len(self._rollout_bounds) == 2:
            raise ShapeErr(msg='The StepSequence must be a single continuous rollout.')
        if not 0 <= gamma <= 1:
            raise ValueErr(given=gamma, ge_constraint='0', le_constraint='1')
        if self._data_format == 'torch':
            return to.dot(self.rewards, (gamma**to.arange(self.length)).to(self.rewards.dtype))
        return np.dot(self.rewards, gamma**np.arange(self.length))

    @classmethod
    def concat(cls, parts, data_format: str = None, truncate_last: bool = True):
        data_format = data_format or parts[0].data_format

        def cat(entries):
            if isinstance(entries[0], dict):
                return {k: cat([e[k] for e in entries]) for k in entries[0]}
            entries = [_convert(e, data_format) for e in entries]
            return to.cat(entries) if data_format == 'torch' else np.concatenate(entries)

        data = {name: cat([ro.get_data_values(name, truncate_last) for ro in parts]) for name in parts[0].data_names}
        done = np.concatenate([ro.done for ro in parts])
        cont = all(ro.continuous for ro in parts)
        bounds = None
        if cont:
            bounds, acc = [0], 0
            for ro in parts:
                bounds.extend(int(b) + acc for b in ro.rollout_bounds[1:])
                acc += int(ro.rollout_bounds[-1])
        return StepSequence(data_format=data_format, done=done, continuous=cont, rollout_bounds=bounds, **data)

    def __deepcopy__(self, memo):
        new = StepSequence.__new__(StepSequence)
        new.__dict__ = {k: deepcopy(v, memo) for k, v in self.__dict__.items()}
        return new


def discounted_reverse_cumsum(data, gamma: float):
    """ y[i] = sum_{j>=i} gamma^(j-i) data[j] """
    out = np.zeros(len(data)) if not isinstance(data, to.Tensor) else to.zeros_like(data)
    acc = 0.
    for i in reversed(range(len(data))):
        acc = data[i] + gamma*acc
        out[i] = acc
    return out


def discounted_value(rollout: StepSequence, gamma: float):
    """ Discounted return from every step to the end of its rollout. """
    parts = [discounted_reverse_cumsum(ro.rewards, gamma) for ro in rollout.iterate_rollouts()]
    return to.cat(parts) if rollout.data_format == 'torch' else np.concatenate(parts)
