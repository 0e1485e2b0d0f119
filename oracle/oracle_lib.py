"""
TEST INFRASTRUCTURE: ctypes loader of the C oracle (oracle/libpysim_oracle.so, built by oracle/Makefile).
Takes NumPy arrays where the product takes device tensors.  Only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from simurlacra_amd import _abi as abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(['make', '-C', _HERE, '-s'])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, 'libpysim_oracle.so')
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_rollout.restype = C.c_int
        L.orc_rollout.argtypes = [C.POINTER(abi.PshRolloutDesc), C.c_int]
        L.orc_reset.restype = C.c_int
        L.orc_reset.argtypes = [C.POINTER(abi.PshStepDesc)]
        L.orc_step.restype = C.c_int
        L.orc_step.argtypes = [C.POINTER(abi.PshStepDesc)]
        L.orc_randomize.restype = C.c_int
        L.orc_randomize.argtypes = [C.POINTER(abi.PshRandDesc)]
        L.orc_sample_init.restype = C.c_int
        L.orc_sample_init.argtypes = [C.POINTER(abi.PshInitDesc)]
        L.orc_qbb_ik.restype = C.c_double
        L.orc_qbb_ik.argtypes = [C.c_double, C.c_double, C.c_double]
        L.orc_qbb_ik_n.restype = C.c_double
        L.orc_qbb_ik_n.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int, abi.c_double_p]
        L.orc_step_rew.restype = C.c_double
        L.orc_step_rew.argtypes = [C.c_int, C.POINTER(abi.PshTask), abi.c_double_p, abi.c_double_p, abi.c_double_p]
        L.orc_constants.restype = None
        L.orc_constants.argtypes = [C.c_int] + [abi.c_double_p]*6
        L.orc_policy_forward.restype = C.c_int
        L.orc_policy_forward.argtypes = [C.POINTER(abi.PshPolicy), C.c_int, C.c_int, abi.c_double_p, abi.c_double_p]
        L.orc_philox.restype = None
        L.orc_philox.argtypes = [C.POINTER(C.c_uint32*4), C.POINTER(C.c_uint32*2), C.POINTER(C.c_uint32*4)]
        L.orc_bootstrap_means.restype = None
        L.orc_bootstrap_means.argtypes = [C.c_int, C.c_int, abi.c_double_p, C.c_uint64, abi.c_double_p]
        L.orc_num_threads.restype = C.c_int
        _LIB = L
    return _LIB


def _ptr(a):
    return 0 if a is None else a.ctypes.data


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def rollout(env_kind, dt, max_steps, task_c, policy_spec, params, init_state, domain_param, *, flags=abi.F_STOP_ON_DONE,
            dp_index=None, policy_index=None, policy_group=0, record=False, wrap=None, bernoulli_reset=0.,
            n_threads=0):
    """ N rollouts.  init_state [init_dim, N], domain_param [rows, M], params [n_policies, num_param]. """
    S, O, A, P, _ = abi.ENV_DIMS[env_kind]
    O_rec = O if wrap is None else wrap.obs_dim(O)
    init_state, domain_param, params = _f64(init_state), _f64(domain_param), np.atleast_2d(_f64(params))
    N = init_state.shape[1]
    T = int(max_steps)
    dp_index = None if dp_index is None else np.ascontiguousarray(dp_index, dtype=np.int32)
    policy_index = None if policy_index is None else np.ascontiguousarray(policy_index, dtype=np.int32)
    out = dict(ret=np.zeros(N), length=np.zeros(N, dtype=np.int32), status=np.zeros(N, dtype=np.int32),
               final_state=np.zeros((S, N)))
    if record:
        out.update(obs=np.full((T + 1, O_rec, N), np.nan), act=np.full((T, A, N), np.nan), rew=np.full((T, N), np.nan))
    d = abi.PshRolloutDesc()
    d.env_kind, d.n_envs, d.max_steps, d.flags, d.dt = env_kind, N, T, flags, dt
    d.init_state, d.init_dim, d.ld_init = _ptr(init_state), init_state.shape[0], N
    d.domain_param, d.dp_index, d.ld_dp = _ptr(domain_param), _ptr(dp_index), domain_param.shape[1]
    d.dp_rows = domain_param.shape[0]
    if wrap is not None:
        d.wrap = wrap.to_c(O, A)
    d.bernoulli_reset = float(bernoulli_reset)
    d.task = task_c
    feat_data = _f64(policy_spec.feat_data()) if policy_spec.kind == 'linear' else None
    d.policy = policy_spec.to_c(params.shape[0], _ptr(params), _ptr(policy_index), policy_group,
                                _ptr(feat_data) if feat_data is not None and feat_data.size else 0)
    d.ret, d.length, d.status, d.final_state = _ptr(out['ret']), _ptr(out['length']), _ptr(out['status']), _ptr(out['final_state'])
    if record:
        d.obs, d.act, d.rew = _ptr(out['obs']), _ptr(out['act']), _ptr(out['rew'])
    rc = lib().orc_rollout(C.byref(d), n_threads)
    if rc != 0:
        raise RuntimeError(f'orc_rollout failed with {rc}')
    return out


def _step_desc(env_kind, dt, max_steps, task_c, domain_param, flags, dp_index):
    d = abi.PshStepDesc()
    d.env_kind, d.max_steps, d.flags, d.dt = env_kind, int(max_steps), flags, dt
    d.domain_param, d.dp_index, d.ld_dp = _ptr(domain_param), _ptr(dp_index), domain_param.shape[1]
    d.task = task_c
    return d


def reset(env_kind, dt, max_steps, task_c, init_state, domain_param, *, flags=0, dp_index=None):
    S, O, A, P, _ = abi.ENV_DIMS[env_kind]
    init_state, domain_param = _f64(init_state), _f64(domain_param)
    N = init_state.shape[1]
    mem = dict(state=np.zeros((S + (2 if env_kind == abi.ENV_QBB else 0), N)), curr_step=np.zeros(N, dtype=np.int32),
               flags_io=np.zeros(N, dtype=np.int32), obs=np.zeros((O, N)), status=np.zeros(N, dtype=np.int32))
    d = _step_desc(env_kind, dt, max_steps, task_c, domain_param, flags, dp_index)
    d.n_envs, d.init_dim, d.init_state = N, init_state.shape[0], _ptr(init_state)
    d.state, d.curr_step, d.flags_io, d.obs, d.status = (_ptr(mem[k]) for k in ('state', 'curr_step', 'flags_io', 'obs', 'status'))
    rc = lib().orc_reset(C.byref(d))
    if rc != 0:
        raise RuntimeError(f'orc_reset failed with {rc}')
    return mem


def step(env_kind, dt, max_steps, task_c, mem, act, domain_param, *, flags=0, dp_index=None):
    """ Advances `mem` (from reset) in place by one step; returns obs, rew, done, status, act (as left by the env). """
    S, O, A, P, _ = abi.ENV_DIMS[env_kind]
    domain_param = _f64(domain_param)
    act = _f64(act).copy()
    N = act.shape[1]
    rew, done = np.zeros(N), np.zeros(N, dtype=np.int32)
    d = _step_desc(env_kind, dt, max_steps, task_c, domain_param, flags, dp_index)
    d.n_envs = N
    d.state, d.curr_step, d.flags_io, d.obs, d.status = (_ptr(mem[k]) for k in ('state', 'curr_step', 'flags_io', 'obs', 'status'))
    d.act, d.rew, d.done = _ptr(act), _ptr(rew), _ptr(done)
    rc = lib().orc_step(C.byref(d))
    if rc != 0:
        raise RuntimeError(f'orc_step failed with {rc}')
    return mem['obs'].copy(), rew, done, mem['status'].copy(), act


def fill_rand_desc(d, distrs, nominal_rows, seed, env_offset):
    d.n_rows, d.n_distr, d.seed, d.env_offset = len(nominal_rows), len(distrs), seed, env_offset
    for i, v in enumerate(nominal_rows):
        d.nominal[i] = float(v)
    for k, q in enumerate(distrs):
        e = d.distr[k]
        e.kind, e.row, e.roundint = q['kind'], q['row'], int(q.get('roundint', 0))
        e.a, e.b, e.c = float(q.get('a', 0.)), float(q.get('b', 0.)), float(q.get('c', 0.))
        e.clip_lo, e.clip_up = float(q.get('clip_lo', -np.inf)), float(q.get('clip_up', np.inf))


def randomize(n_envs, distrs, nominal_rows, seed, env_offset=0):
    out = np.zeros((len(nominal_rows), n_envs))
    d = abi.PshRandDesc()
    d.n_envs, d.ld_out, d.out = n_envs, n_envs, _ptr(out)
    fill_rand_desc(d, distrs, nominal_rows, seed, env_offset)
    rc = lib().orc_randomize(C.byref(d))
    if rc != 0:
        raise RuntimeError(f'orc_randomize failed with {rc}')
    return out


def sample_init(env_kind, n_envs, domain_param, seed, env_offset=0, dp_index=None):
    domain_param = _f64(domain_param)
    out = np.zeros((abi.ENV_DIMS[env_kind][4], n_envs))
    d = abi.PshInitDesc()
    d.env_kind, d.n_envs, d.ld_dp, d.ld_out, d.seed, d.env_offset = env_kind, n_envs, domain_param.shape[1], n_envs, seed, env_offset
    d.domain_param, d.dp_index, d.out = _ptr(domain_param), _ptr(dp_index), _ptr(out)
    rc = lib().orc_sample_init(C.byref(d))
    if rc != 0:
        raise RuntimeError(f'orc_sample_init failed with {rc}')
    return out


def qbb_ik(th, r_arm, l_plate):
    return lib().orc_qbb_ik(float(th), float(r_arm), float(l_plate))


def qbb_ik_n(th, r_arm, l_plate, n_iter):
    """ plate angle and rod tip after n_iter optimizer steps """
    tip = np.zeros(2)
    ang = lib().orc_qbb_ik_n(float(th), float(r_arm), float(l_plate), int(n_iter), tip.ctypes.data_as(abi.c_double_p))
    return ang, tip


def step_rew(env_kind, task_c, dp, state, act):
    dp, state, act = _f64(dp), _f64(state), _f64(act)
    f = lambda a: a.ctypes.data_as(abi.c_double_p)
    return lib().orc_step_rew(env_kind, C.byref(task_c), f(dp), f(state), f(act))


def constants(env_kind, dp):
    S, O, A, P, _ = abi.ENV_DIMS[env_kind]
    dp = _f64(dp)
    c, slo, shi, alo, ahi = np.zeros(8), np.zeros(S), np.zeros(S), np.zeros(A), np.zeros(A)
    f = lambda a: a.ctypes.data_as(abi.c_double_p)
    lib().orc_constants(env_kind, f(dp), f(c), f(slo), f(shi), f(alo), f(ahi))
    return c, slo, shi, alo, ahi


def policy_forward(policy_spec, params, obs):
    params, obs = _f64(params), _f64(obs)
    act = np.zeros(policy_spec.act_dim)
    feat_data = _f64(policy_spec.feat_data()) if policy_spec.kind == 'linear' else None
    pc = policy_spec.to_c(1, _ptr(params), 0, 0, _ptr(feat_data) if feat_data is not None and feat_data.size else 0)
    f = lambda a: a.ctypes.data_as(abi.c_double_p)
    lib().orc_policy_forward(C.byref(pc), policy_spec.obs_dim, policy_spec.act_dim, f(obs), f(act))
    return act


def philox(ctr, key):
    c = (C.c_uint32*4)(*ctr)
    k = (C.c_uint32*2)(*key)
    o = (C.c_uint32*4)()
    lib().orc_philox(C.byref(c), C.byref(k), C.byref(o))
    return [int(v) for v in o]


def bootstrap_means(data, num_reps, seed):
    data = _f64(data).reshape(-1)
    out = np.zeros(int(num_reps))
    lib().orc_bootstrap_means(data.size, int(num_reps), data.ctypes.data_as(abi.c_double_p), int(seed),
                              out.ctypes.data_as(abi.c_double_p))
    return out
