"""
SimPyEnv: base of the standalone simulations (Pyrado/pyrado/environments/pysim/base.py:13-243).

The object holds what the reference object holds on the host (domain parameters, spaces, task, current state); the
numerics of reset() and step() run in the HIP kernels behind psh_reset / psh_step with a batch of one environment.
Batched rollouts do not go through these methods: the samplers hand the same description to psh_rollout.
"""
from copy import deepcopy

import numpy as np
import torch

from ... import _abi as abi, inf
from ...envdefs import DP_NAMES
from ...tasks import task_to_spec
from ...utils.exceptions import TypeErr, ValueErr
from ..sim_base import SimEnv


class SimPyEnv(SimEnv):
    kind: int = None  # environment kind of the C ABI

    def __init__(self, dt: float, max_steps: int = inf, task_args: dict = None):
        super().__init__(dt, max_steps)
        self._domain_param = self.get_nominal_domain_param()
        self._state_space = self._obs_space = self._act_space = self._init_space = None
        self._calc_constants()
        self._create_spaces()
        if not (isinstance(task_args, dict) or task_args is None):
            raise TypeErr(given=task_args, expected_type=dict)
        self._task = self._create_task(dict() if task_args is None else task_args)
        self._curr_act = np.zeros(self.act_space.shape)
        self._yielded = 0      # "final reward already given" flag of the device step
        self._hidden = np.zeros(self._num_hidden())  # extra integrator state that is not part of `state`

    # ------------------------------------------------------------------ static description used by the engine
    @property
    def simplified_dyn(self) -> bool:
        return False

    def _num_hidden(self) -> int:
        return 0

    def engine_flags(self) -> int:
        return abi.F_SIMPLIFIED_DYN if self.simplified_dyn else 0

    def task_spec(self):
        return task_to_spec(self._task)

    def domain_param_column(self, param: dict = None) -> np.ndarray:
        """ The domain parameters as one column of the [P, M] device table (row order of the C ABI). """
        p = self._domain_param if param is None else {**self._domain_param, **param}
        return np.array([float(p[k]) for k in DP_NAMES[self.kind]], dtype=np.float64)

    # ------------------------------------------------------------------------------------------ properties
    @property
    def state_space(self):
        return self._state_space

    @property
    def obs_space(self):
        return self._obs_space

    @property
    def init_space(self):
        return self._init_space

    @property
    def act_space(self):
        return self._act_space

    @property
    def task(self):
        return self._task

    @property
    def domain_param(self) -> dict:
        return deepcopy(self._domain_param)

    @domain_param.setter
    def domain_param(self, param: dict):
        if not isinstance(param, dict):
            raise TypeErr(given=param, expected_type=dict)
        self._domain_param.update(param)
        self._calc_constants()
        self._create_spaces()
        self._task.reset(env_spec=self.spec)

    def _calc_constants(self):
        pass

    def _create_spaces(self):
        raise NotImplementedError

    def _create_task(self, task_args: dict):
        raise NotImplementedError

    def _state_from_init(self, init_state: np.ndarray) -> np.ndarray:
        return init_state

    # ------------------------------------------------------------------------------------- device round trip
    def _engine(self):
        from ...engine import get_engine
        return get_engine()

    def _device_call(self, reset: bool, init_state=None, act=None):
        eng = self._engine()
        S, O, A, P, _ = abi.ENV_DIMS[self.kind]
        X = self._num_hidden()
        dev = eng.device
        dp = torch.as_tensor(self.domain_param_column(), device=dev).reshape(P, 1)
        task_c = self.task_spec().to_c(S, A)
        inf_steps = self._max_steps == inf
        T = 1 if inf_steps else int(self._max_steps)
        flags = self.engine_flags() | (abi.F_MAX_STEPS_INF if inf_steps else 0)
        if reset:
            mem = eng.reset(self.kind, self._dt, T, task_c, np.asarray(init_state, dtype=np.float64).reshape(-1, 1), dp,
                            flags=flags)
            st = mem['state'].cpu().numpy().reshape(-1)
            self.state, self._hidden = st[:S].copy(), st[S:S + X].copy()
            self._yielded = 0
            status = int(mem['status'].item())
            obs = mem['obs'].cpu().numpy().reshape(-1)
            if status & abi.ST_NAN:
                raise ValueErr(msg=f'At least one value is NaN! state {self.state}')
            return obs
        host_state = np.concatenate([np.asarray(self.state, dtype=np.float64).reshape(-1), self._hidden])
        mem = dict(state=torch.as_tensor(host_state, device=dev).reshape(S + X, 1).contiguous(),
                   curr_step=torch.tensor([self._curr_step], dtype=torch.int32, device=dev),
                   flags_io=torch.tensor([self._yielded], dtype=torch.int32, device=dev),
                   obs=torch.empty((O, 1), dtype=torch.float64, device=dev),
                   status=torch.zeros(1, dtype=torch.int32, device=dev))
        act_t = torch.as_tensor(np.asarray(act, dtype=np.float64).reshape(A, 1), device=dev).contiguous()
        obs, rew, done, status = eng.step(self.kind, self._dt, T, task_c, mem, act_t, dp, flags=flags)
        st = mem['state'].cpu().numpy().reshape(-1)
        self.state, self._hidden = st[:S].copy(), st[S:S + X].copy()
        self._yielded = int(mem['flags_io'].item())
        if int(status.item()) & abi.ST_NAN:
            raise ValueErr(msg=f'At least one value is NaN! state {self.state}, action {act}')
        return obs.cpu().numpy().reshape(-1), float(rew.item()), bool(done.item()), act_t.cpu().numpy().reshape(-1)

    # ----------------------------------------------------------------------------------------- reset / step
    def reset(self, init_state: np.ndarray = None, domain_param: dict = None) -> np.ndarray:
        self._curr_step = 0
        if domain_param is not None:
            self.domain_param = domain_param
        if init_state is None:
            init_state = self.init_space.sample_uniform()
        elif not isinstance(init_state, np.ndarray):
            try:
                init_state = np.array(init_state, dtype=np.float64)
            except Exception:
                raise TypeErr(given=init_state, expected_type=[np.ndarray, list])
        self._task.reset(env_spec=self.spec)
        return self._device_call(True, init_state=init_state)

    def step(self, act: np.ndarray) -> tuple:
        obs, rew, done, act_after = self._device_call(False, act=act)
        if isinstance(act, np.ndarray) and act.dtype == np.float64 and act.shape == act_after.shape:
            act[:] = act_after  # the reference's dynamics may write into the caller's action array
        self._curr_act = act_after
        info = dict(t=self._curr_step*self._dt)
        self._curr_step += 1
        self._curr_rew = rew
        return obs, rew, done, info

    def __getstate__(self):
        return self.__dict__.copy()
