"""
Thin Python binding of libpysimhip.so (C ABI: include/pysimhip.h).  PyTorch-ROCm tensors provide device memory
and streams; every numeric result comes from the HIP kernels.  There is no CPU fallback: without the built library
or without a GPU the calls raise.
"""
import ctypes as C
import os
import threading

import numpy as np
import torch

from . import _abi as abi

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc', 'libpysimhip.so')
_lib = None
_lib_lock = threading.Lock()


class EngineError(RuntimeError):
    pass


def load_library():
    """ Load libpysimhip.so and declare the prototypes of every exported symbol. """
    global _lib
    with _lib_lock:
        if _lib is None:
            if not os.path.exists(_LIB_PATH):
                raise EngineError(f'{_LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; '
                                  f'g.build()"` or `make -C simurlacra_amd/csrc`. There is no CPU fallback.')
            lib = C.CDLL(_LIB_PATH)
            for name, (restype, argtypes) in abi.PSH_SYMBOLS.items():
                fn = getattr(lib, name)  # raises AttributeError if the library lacks a declared symbol
                fn.restype = restype
                fn.argtypes = argtypes
            if lib.psh_version() != abi.PSH_VERSION:
                raise EngineError(f'libpysimhip.so has ABI version {lib.psh_version()}, expected {abi.PSH_VERSION}')
            _lib = lib
    return _lib


def _ptr(t):
    return 0 if t is None else t.data_ptr()


class Engine:
    """ One context on one GPU.  All tensor arguments must live on that GPU. """

    def __init__(self, device: int = 0):
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise EngineError('no GPU visible to PyTorch-ROCm: the rollout engine has no CPU fallback')
        self.device = torch.device('cuda', device)
        h = C.c_void_p()
        rc = self.lib.psh_create(device, C.byref(h))
        if rc != 0:
            raise EngineError(f'psh_create({device}) failed with {rc}')
        self._h = h
        self._lock = threading.Lock()

    def close(self):
        if getattr(self, '_h', None):
            self.lib.psh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers
    def _call(self, what, *args):
        """ One library call on this context.  ctypes releases the GIL for the duration of the call and the context's
        host-side state (scratch buffers, error message) is not thread safe, so the calls of one Engine are serialised. """
        with self._lock:
            rc = getattr(self.lib, what)(self._h, *args)
            if rc != 0:
                msg = self.lib.psh_last_error(self._h)
                raise EngineError(f'{what} failed ({rc}): {msg.decode() if msg else ""}')

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def f64(self, a, shape=None):
        t = torch.as_tensor(np.asarray(a, dtype=np.float64) if not torch.is_tensor(a) else a, dtype=torch.float64)
        t = t.to(self.device).contiguous()
        if shape is not None:
            t = t.reshape(shape)
        return t

    def i32(self, a):
        if a is None:
            return None
        t = torch.as_tensor(np.asarray(a, dtype=np.int32) if not torch.is_tensor(a) else a, dtype=torch.int32)
        return t.to(self.device).contiguous()

    # -- fused rollouts
    def rollout(self, env_kind, dt, max_steps, task_c, policy_spec, params, init_state, domain_param, *,
                flags=abi.F_STOP_ON_DONE, dp_index=None, policy_index=None, policy_group=0, record=False,
                wrap=None, bernoulli_reset=0., final_state=True, out=None, static_scratch=False):
        """ N fused rollouts.  init_state [init_dim, N], domain_param [rows, M], params [n_policies, num_param]
        (device tensors, fp64).  Returns a dict of device tensors: ret [N], length [N], status [N], final_state [S, N]
        and, with record=True, obs [T+1, O, N], act [T, A, N], rew [T, N] (with F_TRAJ_ENV_MAJOR among the flags:
        obs [N, T+1, O], act [N, T, A], rew [N, T], every rollout a contiguous block).  `wrap` (a WrapSpec) fuses environment
        wrappers into the step loop; with an ObsPartialWrapper in it O is the number of kept observations.
        policy_group = R > 0: population mode, env i runs params[i // R] (N = n_policies * R).
        static_scratch: issue the launch through psh_rollout_static, which takes its scratch from the library's static
        device array and allocates nothing (pass `out` as well to keep torch from allocating the results). """
        S, O, A, P, _ = abi.ENV_DIMS[env_kind]
        O_rec = O if wrap is None else wrap.obs_dim(O)
        init_state, domain_param = self.f64(init_state), self.f64(domain_param)
        params = self.f64(params)
        if params.dim() == 1:
            params = params.unsqueeze(0)
        dp_index, policy_index = self.i32(dp_index), self.i32(policy_index)
        N, T = init_state.shape[1], int(max_steps)
        dev = self.device
        if out is None:
            out = dict(ret=torch.empty(N, dtype=torch.float64, device=dev),
                       length=torch.full((N,), -1, dtype=torch.int32, device=dev),   # -1: never stored
                       status=torch.empty(N, dtype=torch.int32, device=dev))
            if final_state:
                out['final_state'] = torch.empty((S, N), dtype=torch.float64, device=dev)
            if record:
                em = bool(flags & abi.F_TRAJ_ENV_MAJOR)
                out['obs'] = torch.full((N, T + 1, O_rec) if em else (T + 1, O_rec, N), float('nan'), dtype=torch.float64, device=dev)
                out['act'] = torch.full((N, T, A) if em else (T, A, N), float('nan'), dtype=torch.float64, device=dev)
                out['rew'] = torch.full((N, T) if em else (T, N), float('nan'), dtype=torch.float64, device=dev)
        d = abi.PshRolloutDesc()
        d.env_kind, d.n_envs, d.max_steps, d.flags, d.dt = env_kind, N, T, flags, float(dt)
        d.init_state, d.init_dim, d.ld_init = _ptr(init_state), init_state.shape[0], N
        d.domain_param, d.dp_index, d.ld_dp = _ptr(domain_param), _ptr(dp_index), domain_param.shape[1]
        d.dp_rows = domain_param.shape[0]
        if wrap is not None:
            d.wrap = wrap.to_c(O, A)
        d.bernoulli_reset = float(bernoulli_reset)
        d.task = task_c
        # data of RandFourierFeat / RBFFeat: freed after the launch, which the caching allocator orders on this stream
        feat_data = policy_spec.feat_data() if policy_spec.kind == 'linear' else None
        feat_data = self.f64(feat_data) if feat_data is not None and feat_data.size else None
        d.policy = policy_spec.to_c(params.shape[0], _ptr(params), _ptr(policy_index), policy_group, _ptr(feat_data))
        d.ret, d.length, d.status = _ptr(out['ret']), _ptr(out['length']), _ptr(out['status'])
        d.final_state = _ptr(out.get('final_state'))
        d.obs, d.act, d.rew = _ptr(out.get('obs')), _ptr(out.get('act')), _ptr(out.get('rew'))
        d.stream = self._stream()
        if static_scratch:
            self._call('psh_rollout_static', C.byref(d))
        else:
            self._call('psh_rollout', C.byref(d))
        return out

    def last_rollout_ms(self) -> float:
        return float(self.lib.psh_last_rollout_ms(self._h))

    def rollout_ms(self, back: int = 0) -> float:
        """ Kernel time (HIP events on the launch stream) of the rollout issued `back` launches before the latest one. """
        return float(self.lib.psh_rollout_ms(self._h, int(back)))

    def alloc_count(self) -> int:
        """ Device allocations the library has made for this context's scratch buffers. """
        return int(self.lib.psh_alloc_count(self._h))

    # -- single steps
    def _step_desc(self, env_kind, dt, max_steps, task_c, domain_param, flags, dp_index):
        d = abi.PshStepDesc()
        d.env_kind, d.max_steps, d.flags, d.dt = env_kind, int(max_steps), flags, float(dt)
        d.domain_param, d.dp_index, d.ld_dp = _ptr(domain_param), _ptr(dp_index), domain_param.shape[1]
        d.task = task_c
        d.stream = self._stream()
        return d

    def reset(self, env_kind, dt, max_steps, task_c, init_state, domain_param, *, flags=0, dp_index=None):
        """ SimPyEnv.reset for N envs.  Returns the environment memory (dict of device tensors). """
        S, O, A, P, _ = abi.ENV_DIMS[env_kind]
        init_state, domain_param, dp_index = self.f64(init_state), self.f64(domain_param), self.i32(dp_index)
        N, dev = init_state.shape[1], self.device
        X = 2 if env_kind == abi.ENV_QBB else 0
        mem = dict(state=torch.zeros((S + X, N), dtype=torch.float64, device=dev),
                   curr_step=torch.zeros(N, dtype=torch.int32, device=dev),
                   flags_io=torch.zeros(N, dtype=torch.int32, device=dev),
                   obs=torch.zeros((O, N), dtype=torch.float64, device=dev),
                   status=torch.zeros(N, dtype=torch.int32, device=dev))
        d = self._step_desc(env_kind, dt, max_steps, task_c, domain_param, flags, dp_index)
        d.n_envs, d.init_dim, d.init_state = N, init_state.shape[0], _ptr(init_state)
        d.state, d.curr_step, d.flags_io, d.obs, d.status = (_ptr(mem[k]) for k in ('state', 'curr_step', 'flags_io', 'obs', 'status'))
        self._call('psh_reset', C.byref(d))
        return mem

    def step(self, env_kind, dt, max_steps, task_c, mem, act, domain_param, *, flags=0, dp_index=None):
        """ SimPyEnv.step for N envs, advancing `mem` in place.  `act` [A, N] is updated in place as the reference
        updates the caller's action array (QBB dead zone).  Returns obs [O, N], rew [N], done [N], status [N]. """
        domain_param, dp_index = self.f64(domain_param), self.i32(dp_index)
        N, dev = act.shape[1], self.device
        rew = torch.empty(N, dtype=torch.float64, device=dev)
        done = torch.empty(N, dtype=torch.int32, device=dev)
        d = self._step_desc(env_kind, dt, max_steps, task_c, domain_param, flags, dp_index)
        d.n_envs = N
        d.state, d.curr_step, d.flags_io, d.obs, d.status = (_ptr(mem[k]) for k in ('state', 'curr_step', 'flags_io', 'obs', 'status'))
        d.act, d.rew, d.done = _ptr(act), _ptr(rew), _ptr(done)
        self._call('psh_step', C.byref(d))
        return mem['obs'], rew, done, mem['status']

    # -- randomization
    def randomize(self, n_envs, distrs, nominal_rows, seed, env_offset=0):
        """ [rows, n_envs] table of domain parameters drawn on the device.  `distrs`: list of dicts with keys
        kind,row,a,b,c,clip_lo,clip_up,roundint; rows without a distribution hold nominal_rows[row]. """
        if len(distrs) > abi.PSH_MAX_DISTR:
            raise ValueError(f'at most {abi.PSH_MAX_DISTR} distributions per randomizer')
        out = torch.empty((len(nominal_rows), n_envs), dtype=torch.float64, device=self.device)
        d = abi.PshRandDesc()
        d.n_envs, d.ld_out, d.out, d.stream = n_envs, n_envs, _ptr(out), self._stream()
        d.n_rows, d.n_distr, d.seed, d.env_offset = len(nominal_rows), len(distrs), int(seed), int(env_offset)
        for i, v in enumerate(nominal_rows):
            d.nominal[i] = float(v)
        for k, q in enumerate(distrs):
            e = d.distr[k]
            e.kind, e.row, e.roundint = q['kind'], q['row'], int(q.get('roundint', 0))
            e.a, e.b, e.c = float(q.get('a', 0.)), float(q.get('b', 0.)), float(q.get('c', 0.))
            e.clip_lo, e.clip_up = float(q.get('clip_lo', -np.inf)), float(q.get('clip_up', np.inf))
        self._call('psh_randomize', C.byref(d))
        return out

    def sample_init(self, env_kind, n_envs, domain_param, seed, env_offset=0, dp_index=None):
        domain_param, dp_index = self.f64(domain_param), self.i32(dp_index)
        out = torch.empty((abi.ENV_DIMS[env_kind][4], n_envs), dtype=torch.float64, device=self.device)
        d = abi.PshInitDesc()
        d.env_kind, d.n_envs, d.ld_dp, d.ld_out = env_kind, n_envs, domain_param.shape[1], n_envs
        d.seed, d.env_offset = int(seed), int(env_offset)
        d.domain_param, d.dp_index, d.out, d.stream = _ptr(domain_param), _ptr(dp_index), _ptr(out), self._stream()
        self._call('psh_sample_init', C.byref(d))
        return out

    def qbb_ik(self, th, r_arm, l_plate):
        th, r_arm, l_plate = self.f64(th).reshape(-1), self.f64(r_arm).reshape(-1), self.f64(l_plate).reshape(-1)
        ang = torch.empty_like(th)
        self._call('psh_qbb_ik', th.numel(), _ptr(th), _ptr(r_arm), _ptr(l_plate), _ptr(ang), self._stream())
        return ang

    def eval_math(self, fn: str, x, m=None):
        """ Element-wise device math (diagnostic): fn in exp, tanh, sigmoid, sin, cos, fmod, log. """
        ids = dict(exp=0, tanh=1, sigmoid=2, sin=3, cos=4, fmod=5, tanh_table_fine=6, tanh_table_coarse=7, log=8)
        x = self.f64(x).reshape(-1)
        m = None if m is None else self.f64(m).reshape(-1)
        y = torch.empty_like(x)
        self._call('psh_eval_math', ids[fn], x.numel(), _ptr(x), _ptr(m), _ptr(y), self._stream())
        return y

    def segment_mean(self, ret, group_size):
        ret = self.f64(ret).reshape(-1)
        n_groups = ret.numel()//group_size
        mean = torch.empty(n_groups, dtype=torch.float64, device=self.device)
        self._call('psh_segment_mean', n_groups, group_size, _ptr(ret), _ptr(mean), self._stream())
        return mean

    def topk(self, values, k: int):
        """ Indices (int32 [k]) and values ([k]) of the k largest entries in descending order, on the device. """
        values = self.f64(values).reshape(-1)
        idx = torch.empty(k, dtype=torch.int32, device=self.device)
        val = torch.empty(k, dtype=torch.float64, device=self.device)
        self._call('psh_topk', values.numel(), int(k), _ptr(values), _ptr(idx), _ptr(val), self._stream())
        return idx, val

    def bootstrap_means(self, data, num_reps: int, seed: int):
        """ [num_reps] means of len(data) draws with replacement each, on the device. """
        data = self.f64(data).reshape(-1)
        out = torch.empty(int(num_reps), dtype=torch.float64, device=self.device)
        self._call('psh_bootstrap_means', data.numel(), int(num_reps), _ptr(data), int(seed), _ptr(out),
                   self._stream())
        return out


_engines = {}


def get_engine(device: int = None) -> Engine:
    """ Process-wide engine per GPU (created on first use). """
    if device is None:
        device = torch.cuda.current_device() if torch.cuda.is_available() else 0
    if device not in _engines:
        _engines[device] = Engine(device)
    return _engines[device]
