"""
Batched rollouts: what the reference does with one rollout() call per (env copy, policy copy) in worker processes
(Pyrado/pyrado/sampling/sampler_pool.py:286-410, parallel_sampler.py:104-123) is here ONE launch of the fused
rollout kernel over N environment lanes.  This module turns an environment chain and a policy into the arguments of
psh_rollout (or, for policies the kernels cannot fuse, into a psh_reset / psh_step loop with the policy evaluated by
torch on the whole batch) and wraps the result so that it can be consumed like the reference's List[StepSequence].
"""
from collections.abc import Sequence

import numpy as np
import torch

from .. import _abi as abi, inf
from ..envdefs import DP_NAMES
from ..environment_wrappers.action_delay import ActDelayWrapper
from ..environment_wrappers.action_noise import GaussianActNoiseWrapper
from ..environment_wrappers.action_normalization import ActNormWrapper
from ..environment_wrappers.base import EnvWrapper
from ..environment_wrappers.observation_noise import GaussianObsNoiseWrapper
from ..environment_wrappers.observation_normalization import ObsNormWrapper
from ..environment_wrappers.observation_partial import ObsPartialWrapper
from ..environment_wrappers.domain_randomization import (DomainRandWrapper, DomainRandWrapperBuffer,
                                                          DomainRandWrapperLive, MetaDomainRandWrapper)
from ..environment_wrappers.utils import all_envs, inner_env
from ..environments.pysim.base import SimPyEnv
from ..specs import WrapSpec
from ..utils.exceptions import TypeErr, ValueErr
from .step_sequence import StepSequence


_WRAP_OPS = {ActNormWrapper: 'act_norm', ActDelayWrapper: 'act_delay', GaussianActNoiseWrapper: 'act_noise',
             GaussianObsNoiseWrapper: 'obs_noise', ObsNormWrapper: 'obs_norm', ObsPartialWrapper: 'obs_partial'}


class EnvPlan:
    """ Static analysis of an environment chain: the simulation at its core and the wrappers around it. """

    def __init__(self, env):
        self.env = env
        self.sim = inner_env(env)
        if not isinstance(self.sim, SimPyEnv):
            raise NotImplementedError(f'{type(self.sim).__name__} is not one of the simulations the rollout kernels '
                                      f'implement (QQube, QQubeStab, QCartPoleSwingUp, QCartPoleStab, QBallBalancer, BallOnBeam, '
                                      f'OneMassOscillator, Pendulum)')
        self.kind = self.sim.kind
        self.S, self.O, self.A, self.P, self.I = abi.ENV_DIMS[self.kind]
        self.dr_wrapper = None       # outermost DomainRandWrapperLive/Buffer (Meta wrappers forward to it)
        self.wrappers = []           # action / observation wrappers from the outside in: (op name, wrapper)
        for e in all_envs(env):
            if isinstance(e, MetaDomainRandWrapper):
                continue
            if isinstance(e, DomainRandWrapper):
                if self.dr_wrapper is None:
                    self.dr_wrapper = e
            elif type(e) in _WRAP_OPS:
                op = _WRAP_OPS[type(e)]
                if any(op == o for o, _ in self.wrappers):
                    raise NotImplementedError(f'two {type(e).__name__} in one environment chain are not supported by '
                                              f'the batched rollout engine')
                self.wrappers.append((op, e))
            elif isinstance(e, EnvWrapper):
                raise NotImplementedError(f'{type(e).__name__} is not supported by the batched rollout engine')
        self.names = DP_NAMES[self.kind]
        self.delay_wrapper = next((w for op, w in self.wrappers if op == 'act_delay'), None)
        # the delay travels as one more row of the domain-parameter table (it is the domain parameter 'act_delay')
        self.extra_rows = ['act_delay'] if self.delay_wrapper is not None else []

    @property
    def obs_dim(self) -> int:
        """ Entries of the observation as the policy sees it. """
        return int(self.env.obs_space.flat_dim)

    def flags(self, stop_on_done=True) -> int:
        return self.sim.engine_flags() | (abi.F_STOP_ON_DONE if stop_on_done else 0)

    def wrap_spec(self, noise_seed: int = 0, env_offset: int = 0):
        """ The wrapper chain as the kernels take it (vectors indexed by the simulation's observation / action
        index).  A bare simulation gets an empty chain that still carries the key of the noise streams (exploration
        noise and the stochastic reset draw from them). """
        if not self.wrappers:
            return WrapSpec(ops=[], noise_seed=noise_seed, env_offset=env_offset)
        O = self.O
        spec = WrapSpec(ops=[op for op, _ in self.wrappers], noise_seed=noise_seed, env_offset=env_offset)
        partial = next((w for op, w in self.wrappers if op == 'obs_partial'), None)
        keep = np.ones(O, dtype=bool) if partial is None else np.asarray(partial.keep_mask, dtype=bool)

        def to_sim_index(vec, fill):
            """ a vector over the observation of a wrapper's inner env, which may already be the reduced one """
            vec = np.asarray(vec, dtype=np.float64).reshape(-1)
            if vec.size == O:
                return vec
            full = np.full(O, fill, dtype=np.float64)
            full[keep] = vec
            return full
        labels = self.sim.obs_space.labels
        for op, w in self.wrappers:
            if op == 'act_delay':
                spec.act_delay, spec.act_delay_row = int(w.delay), self.P
            elif op == 'act_noise':
                spec.act_noise_mean, spec.act_noise_std = w._mean, w._std
            elif op == 'obs_noise':
                spec.obs_noise_mean, spec.obs_noise_std = to_sim_index(w._mean, 0.), to_sim_index(w._std, 0.)
            elif op == 'obs_partial':
                spec.obs_keep = keep
            elif op == 'obs_norm':
                lo, hi = np.full(O, np.nan), np.full(O, np.nan)
                for r, name in enumerate(labels):
                    if w.explicit_lb and w.explicit_lb.get(name) is not None:
                        lo[r] = w.explicit_lb[name]
                    if w.explicit_ub and w.explicit_ub.get(name) is not None:
                        hi[r] = w.explicit_ub[name]
                w.obs_space  # raises the reference's ValueErr for an infinite bound that is not overridden
                spec.obs_lo, spec.obs_hi = lo, hi
        return spec

    def dp_column(self, d: dict = None) -> np.ndarray:
        """ One column of the device table: the simulation's parameters (overridden by `d`), then the action delay """
        col = self.sim.domain_param_column(d)
        if self.delay_wrapper is not None:
            delay = int(d['act_delay']) if d is not None and 'act_delay' in d else int(self.delay_wrapper.delay)
            col = np.append(col, float(delay))
        return col

    def dp_dicts(self, table: np.ndarray, index=None, extra=None):
        """ Domain-parameter dicts (as rollout_info['domain_param'] holds them) for the columns of a table. """
        base = self.env.domain_param
        cols = range(table.shape[1]) if index is None else index
        out = []
        for c in cols:
            d = dict(base)
            d.update({k: table[r, c] for r, k in enumerate(self.names)})
            if self.extra_rows and table.shape[0] > self.P:
                d['act_delay'] = int(table[self.P, c])
            if extra:
                d.update(extra)
            out.append(d)
        return out


def sample_domain_params(plan: EnvPlan, eng, n: int, seed: int, env_offset: int = 0, n_global: int = None,
                         shard_start: int = 0):
    """ Domain parameters for n environments as a device table [P (+1), M] plus an index [n] (or None when M == n).
    Live wrapper: drawn on the device from the randomizer's distributions (host torch.distributions only for
    distributions without a device sampler); buffer wrapper: the buffer is the table and the ring index advances by
    n; no wrapper: the environment's current parameters for every lane.  With an ActDelayWrapper in the chain the
    table has one more row, the delay of each column (the domain parameter 'act_delay').  Under torch.distributed a
    rank passes the global count of this call as n_global and the index of its first rollout within it as shard_start. """
    w = plan.dr_wrapper
    nominal = plan.sim.domain_param_column()
    delay = [float(plan.delay_wrapper.delay)] if plan.delay_wrapper is not None else []
    column = plan.dp_column
    if w is None:
        return eng.f64(column().reshape(-1, 1)), torch.zeros(n, dtype=torch.int32, device=eng.device)
    if isinstance(w, DomainRandWrapperBuffer):
        if w.buffer is None:
            # DomainRandWrapperBuffer.reset (Pyrado/pyrado/environment_wrappers/domain_randomization.py:179-199) indexes
            # the buffer: an unfilled one is an error there too
            raise TypeErr(msg='The buffer of the DomainRandWrapperBuffer is empty: call fill_buffer() first!')
        buf = w.buffer if isinstance(w.buffer, list) else [w.buffer]
        table = np.stack([column(d) for d in buf], axis=1)
        start = w.ring_idx or 0
        # the ring is indexed by the GLOBAL rollout index and advances by the global count, so every rank count
        # replays the same entries
        total = n if n_global is None else n_global
        idx = (start + shard_start + np.arange(n)) % len(buf)
        w._ring_idx = int((start + total) % len(buf))
        return eng.f64(table), eng.i32(idx)
    rand = w.randomizer
    try:
        distrs = rand.device_table(plan.names, plan.extra_rows)
    except NotImplementedError:
        # no device sampler for one of the distributions: the host draws, from a generator seeded by the call's key,
        # the whole global batch on every rank (identical everywhere), and each rank keeps its own columns
        total = n if n_global is None else n_global
        first = shard_start
        state = torch.random.get_rng_state()
        torch.manual_seed(int(seed) & 0x7fffffffffffffff)
        try:
            rand.randomize(total)
        finally:
            torch.random.set_rng_state(state)
        lst = rand.get_params(-1, format='list', dtype='numpy')
        lst = [lst] if isinstance(lst, dict) else lst
        lst = lst[first:first + n]
        keys = list(plan.names) + plan.extra_rows
        return eng.f64(np.stack([column({k: float(v) for k, v in d.items() if k in keys}) for d in lst], axis=1)), None
    return eng.randomize(n, distrs, list(nominal) + delay, seed=seed, env_offset=env_offset), None


def sample_init_states(plan: EnvPlan, eng, n: int, dp, dp_index, seed: int, env_offset: int = 0):
    return eng.sample_init(plan.kind, n, dp, seed=seed, env_offset=env_offset, dp_index=dp_index)


def run_batch(plan: EnvPlan, eng, policy, init_state, dp, dp_index=None, *, max_steps=None, stop_on_done=True,
              record=True, params=None, policy_index=None, policy_group=0, noise_seed=0, env_offset=0,
              bernoulli_reset=None):
    """ N rollouts on the device.  Fused kernel when the policy allows it (the wrappers of the env chain are fused
    with it), else a step loop with the policy evaluated by torch on the whole batch.  Returns the dict of device
    tensors of Engine.rollout. """
    T = plan.env.max_steps if max_steps is None else max_steps
    if T == inf:
        raise ValueErr(msg='batched rollouts need a finite max_steps')
    T = int(T)
    p_reset = 0. if bernoulli_reset is None else float(bernoulli_reset)
    if not 0. <= p_reset <= 1.:
        raise ValueErr(given=bernoulli_reset, ge_constraint='0', le_constraint='1')
    S, A = plan.S, plan.A
    task_c = plan.sim.task_spec().to_c(S, A)
    spec = policy.fused_spec() if hasattr(policy, 'fused_spec') else None
    wrap = plan.wrap_spec(noise_seed, env_offset)
    if spec is not None:
        if spec.obs_dim != plan.obs_dim:
            raise ValueErr(msg=f'the policy takes {spec.obs_dim} observations, the environment yields {plan.obs_dim}')
        if params is None:
            params = policy.fused_params().detach().to(torch.float64)
        return eng.rollout(plan.kind, plan.sim.dt, T, task_c, spec, params, init_state, dp, flags=plan.flags(stop_on_done),
                           dp_index=dp_index, policy_index=policy_index, policy_group=policy_group, record=record,
                           wrap=wrap, bernoulli_reset=p_reset)
    if params is not None:
        raise NotImplementedError('per-environment policy parameters need a fusable policy')
    return _run_batch_stepwise(plan, eng, policy, init_state, dp, dp_index, T, stop_on_done, record, task_c, wrap,
                               p_reset)


def _obs_bounds(plan, dp, dp_index, N):
    """ Per-lane observation bounds (lo, hi) of the simulation, [O, N] each (they can depend on its domain
    parameters), as the `obs_bounds` of the device-side simulations state them. """
    cols = dp if dp_index is None else dp[:, dp_index.long()]
    if cols.shape[1] != N:
        cols = cols.expand(-1, N)
    k, dev = plan.kind, cols.device
    one = torch.ones((1, N), dtype=torch.float64, device=dev)
    pi, big = float(np.pi), float('inf')
    if k == abi.ENV_QQ:
        hi = torch.cat([one, one, one, one, one*big, one*big])
    elif k == abi.ENV_QQ_ST:       # observes the error to upright, bounded like its state space (asymmetric in alpha)
        lo = torch.cat([one*(-120./180*pi), one*(pi - 15/180.*pi), one*(-20*pi), one*(-20*pi)])
        hi = torch.cat([one*(120./180*pi), one*(pi + 15/180.*pi), one*(20*pi), one*(20*pi)])
        return lo, hi
    elif k in (abi.ENV_QCP_SU, abi.ENV_QCP_ST):
        hi = torch.cat([cols[2:3]/2., one, one, one*big, one*big])
    elif k == abi.ENV_QBB:
        hi = torch.cat([one*pi/4., one*pi/4., cols[3:4]/2., cols[3:4]/2., one*5*pi, one*5*pi, one*0.5, one*0.5])
    elif k == abi.ENV_BOB:
        hi = torch.cat([cols[4:5]/2., one*pi/4., one*10., one*pi])
    elif k == abi.ENV_PEND:
        hi = torch.cat([one, one, one*big])
    elif k == abi.ENV_OMO:
        hi = torch.cat([one, one*10.])
    else:
        raise NotImplementedError(f'observation bounds of environment kind {k}')
    return -hi, hi


def _run_batch_stepwise(plan, eng, policy, init_state, dp, dp_index, T, stop_on_done, record, task_c, wrap=None,
                        p_reset=0.):
    """ rollout() loop (Pyrado/pyrado/sampling/rollout.py:133-254) for N envs at once with an arbitrary torch policy;
    the wrappers of the chain are applied to the whole batch between policy and psh_step.  Noise comes from torch's
    generator here (the fused kernels use their own counter-based one). """
    kind, dt, dev = plan.kind, plan.sim.dt, eng.device
    init_state, dp = eng.f64(init_state), eng.f64(dp)
    dp_index = eng.i32(dp_index)
    N, O, A = init_state.shape[1], plan.O, plan.A
    ops = [] if wrap is None else list(wrap.ops)
    Oe = O if wrap is None else wrap.obs_dim(O)
    flags = plan.sim.engine_flags()
    mem = eng.reset(kind, dt, T, task_c, init_state, dp, flags=flags, dp_index=dp_index)
    out = dict(ret=torch.zeros(N, dtype=torch.float64, device=dev), length=torch.zeros(N, dtype=torch.int32, device=dev),
               status=torch.zeros(N, dtype=torch.int32, device=dev))
    if record:
        out['obs'] = torch.full((T + 1, Oe, N), float('nan'), dtype=torch.float64, device=dev)
        out['act'] = torch.full((T, A, N), float('nan'), dtype=torch.float64, device=dev)
        out['rew'] = torch.full((T, N), float('nan'), dtype=torch.float64, device=dev)
    live = torch.ones(N, dtype=torch.bool, device=dev)
    pol_dev = next((p.device for p in policy.parameters()), torch.device('cpu')) if hasattr(policy, 'parameters') else torch.device('cpu')

    def vec(v, n):
        return torch.as_tensor(np.zeros(n) if v is None else np.asarray(v, dtype=np.float64), device=dev).reshape(n, 1)
    keep = None
    if 'obs_partial' in ops:
        keep = torch.as_tensor(np.asarray(wrap.obs_keep, dtype=bool), device=dev)
    if 'obs_norm' in ops:
        o_lo, o_hi = _obs_bounds(plan, dp, dp_index, N)
        lo_ov, hi_ov = vec(wrap.obs_lo, O), vec(wrap.obs_hi, O)
        o_lo = torch.where(torch.isnan(lo_ov), o_lo, lo_ov)
        o_hi = torch.where(torch.isnan(hi_ov), o_hi, hi_ov)

    def process_obs(o):  # observation wrappers, innermost first
        for op in reversed(ops):
            if op == 'obs_noise':
                o = o + (torch.randn((O, N), dtype=torch.float64, device=dev)*vec(wrap.obs_noise_std, O)
                         + vec(wrap.obs_noise_mean, O))
            elif op == 'obs_norm':
                o = (o - o_lo)/(o_hi - o_lo)*2 - 1
        return o if keep is None else o[keep]
    # ActDelayWrapper with per-lane delays: a history ring of the last PSH_MAX_DELAY + 1 actions
    delays = hist = None
    if 'act_delay' in ops:
        cols = dp if dp_index is None else dp[:, dp_index.long()]
        delays = (cols[wrap.act_delay_row] if wrap.act_delay_row >= 0 else torch.full((N,), float(wrap.act_delay), device=dev))
        delays = delays.expand(N).clamp(0, abi.PSH_MAX_DELAY).long()
        H = int(delays.max().item()) + 1
        hist = torch.zeros((H, A, N), dtype=torch.float64, device=dev)
    alias = not ('act_norm' in ops or 'act_noise' in ops)
    lane = torch.arange(N, device=dev)
    if hasattr(policy, 'reset'):
        policy.reset()
    k = 0
    obs = process_obs(mem['obs'].clone())
    while bool(live.any()) and k < T:
        if record:
            out['obs'][k][:, live] = obs[:, live]
        with torch.no_grad():
            act = policy(obs.t().to(pol_dev)).to(dev, torch.float64).reshape(N, A).t().contiguous()
        nan = torch.isnan(act).any(dim=0) | torch.isnan(obs).any(dim=0)
        inner = act.clone()
        for op in ops:  # action wrappers, outermost first
            if op == 'act_norm':
                lo, hi = _act_bounds(plan, dp, dp_index, N)
                inner = lo + (inner + 1)*(hi - lo)/2
            elif op == 'act_delay' and hist.shape[0] > 1:
                hist[k % hist.shape[0]] = inner
                src = (k - delays) % hist.shape[0]
                inner = torch.where((k >= delays).unsqueeze(0), hist[src, :, lane].t(), torch.zeros_like(inner))
            elif op == 'act_noise':
                inner = inner + (torch.randn((A, N), dtype=torch.float64, device=dev)*vec(wrap.act_noise_std, A)
                                 + vec(wrap.act_noise_mean, A))
        inner = inner.contiguous()
        mem_prev = {key: v.clone() for key, v in mem.items()}
        o, r, d, st = eng.step(kind, dt, T, task_c, mem, inner, dp, flags=flags, dp_index=dp_index)
        # lanes that were not live keep their memory
        for key in mem:
            sel = ~live
            if mem[key].dim() == 2:
                mem[key][:, sel] = mem_prev[key][:, sel]
            else:
                mem[key][sel] = mem_prev[key][sel]
        stepped = live & ~nan
        if record:
            out['act'][k][:, stepped] = act[:, stepped]
            if alias:
                # the simulation wrote (QBB dead zone) into the array it received: the action recorded `delay` steps ago
                back = (k - delays) if delays is not None else torch.full((N,), k, device=dev)
                sel = stepped & (back >= 0)
                out['act'][back[sel], :, lane[sel]] = inner[:, sel].t()
            out['rew'][k][stepped] = r[stepped]
        out['ret'] += torch.where(stepped, r, torch.zeros_like(r))
        out['length'] += stepped.to(torch.int32)
        out['status'] |= torch.where(stepped, st, torch.zeros_like(st)) | ((live & nan).to(torch.int32)*abi.ST_NAN)
        obs = torch.where(stepped.unsqueeze(0), process_obs(o), obs)
        done = d.bool() & stepped
        live = stepped & ~(done & stop_on_done) & ~((st & abi.ST_NAN) != 0)
        if p_reset > 0. and k + 1 < T:  # stochastic reset (Pyrado/pyrado/sampling/rollout.py:248-254)
            hit = live & (torch.rand(N, dtype=torch.float64, device=dev) < p_reset)
            out['status'] |= hit.to(torch.int32)*abi.ST_BERNOULLI
            live = live & ~hit
        k += 1
    if record:
        idx = out['length'].long()
        out['obs'][idx, :, torch.arange(N, device=dev)] = obs.t()
    S = plan.S
    out['final_state'] = mem['state'][:S].clone()
    return out


def _act_bounds(plan, dp, dp_index, N):
    """ Per-lane action bounds of the inner simulation (they depend on its domain parameters). """
    cols = dp if dp_index is None else dp[:, dp_index.long()]
    k = plan.kind
    if k in (abi.ENV_QQ, abi.ENV_QQ_ST):
        hi = cols[9:10]            # max_volt
    elif k in (abi.ENV_QCP_SU, abi.ENV_QCP_ST):
        hi = torch.full((1, N), 8., dtype=torch.float64, device=cols.device)
    elif k == abi.ENV_QBB:
        hi = torch.full((2, N), 3., dtype=torch.float64, device=cols.device)
    elif k == abi.ENV_BOB:
        hi = cols[4:5]/2.*cols[0:1]*3.
    elif k == abi.ENV_PEND:
        hi = cols[4:5]             # tau_max
    elif k == abi.ENV_OMO:
        hi = cols[1:2]             # k
    else:
        raise NotImplementedError(f'action bounds of environment kind {k}')
    if hi.shape[1] != N:
        hi = hi.expand(-1, N)
    return -hi, hi


class RolloutSummary:
    """ What remains of a rollout when trajectories were not recorded: return, length, rollout_info. """

    def __init__(self, ret, length, rollout_info):
        self._ret, self.length, self.rollout_info = float(ret), int(length), rollout_info

    def undiscounted_return(self) -> float:
        return self._ret

    def __len__(self):
        return self.length


class RolloutBatch(Sequence):
    """ Result of a batched sampling call; behaves like the reference's List[StepSequence].  Items are built on
    access as views into the batched [T(+1), dim, N] arrays, so a million rollouts do not become a million Python
    objects unless someone iterates over all of them.  Results come to the host one array at a time, on first use: the
    per-rollout numbers (returns, lengths, status, domain parameters) as whole arrays, trajectories rollout by rollout
    for the first BULK_AFTER items and as whole arrays from then on (a caller that looks at a few rollouts of a large
    batch does not pay for gigabytes of device-to-host copies; one that iterates over all of them pays once). """
    BULK_AFTER = 32

    def __init__(self, plan: EnvPlan, out: dict, dp, dp_index=None, env_spec=None, extra_dp=None, offset=0, count=None):
        self.plan, self._out = plan, out
        self._dp, self._dp_index = dp, dp_index
        self._host = {}      # name -> numpy array; shared with the slices of this batch
        self._env_spec = env_spec if env_spec is not None else plan.env.spec
        self._extra = extra_dp
        self._offset = offset
        self._count = int(out['ret'].shape[0]) - offset if count is None else count
        self.dt = plan.sim.dt

    def _h(self, key):
        if key not in self._host:
            v = self._dp if key == 'dp' else (self._dp_index if key == 'dp_index' else self._out[key])
            self._host[key] = None if v is None else v.detach().cpu().numpy()
        return self._host[key]

    def _trajectory(self, g: int, L: int):
        """ observations [L + 1, O], actions [L, A], rewards [L] of rollout g as host arrays """
        n = self._host.get('_items', 0)
        if 'obs' not in self._host and n < self.BULK_AFTER:
            self._host['_items'] = n + 1
            o = self._out
            return (o['obs'][:L + 1, :, g].cpu().numpy(), o['act'][:L, :, g].cpu().numpy(), o['rew'][:L, g].cpu().numpy())
        return self._h('obs')[:L + 1, :, g], self._h('act')[:L, :, g], self._h('rew')[:L, g]

    def __len__(self):
        return self._count

    @property
    def device_results(self) -> dict:
        return self._out

    def undiscounted_returns(self) -> np.ndarray:
        return self._h('ret')[self._offset:self._offset + self._count]

    @property
    def lengths(self) -> np.ndarray:
        return self._h('length')[self._offset:self._offset + self._count]

    @property
    def num_steps(self) -> int:
        return int(self.lengths.sum())

    def domain_param(self, i: int) -> dict:
        idx = self._h('dp_index')
        col = i if idx is None else int(idx[i])
        n = self._host.get('_dp_items', 0)
        if 'dp' not in self._host and n < self.BULK_AFTER:   # one column of a table that may hold a million of them
            self._host['_dp_items'] = n + 1
            return self.plan.dp_dicts(self._dp[:, col:col + 1].cpu().numpy(), [0], self._extra)[0]
        return self.plan.dp_dicts(self._h('dp'), [col], self._extra)[0]

    def check(self):
        """ Raise as rollout() does when an observation, action or state became NaN in any lane. """
        st = self._h('status')[self._offset:self._offset + self._count]
        # results the launch never wrote (Engine.rollout pre-fills the lengths with -1) or wrote under an internal fault
        lost = np.flatnonzero(self._h('length')[self._offset:self._offset + self._count] < 0)
        if lost.size:
            raise RuntimeError(f'the rollout kernel stored no result for {lost.size} rollouts (first: {int(lost[0])})')
        if np.any(st & abi.ST_INTERNAL):
            raise RuntimeError(f'the rollout kernel reported an internal fault (status bit {abi.ST_INTERNAL})')
        bad = np.flatnonzero(st & abi.ST_BAD_INDEX)
        if bad.size:
            raise ValueErr(msg=f'Rollout {int(bad[0])} of the batch names a domain-parameter column or a policy outside '
                               f'its table ({bad.size} in total)!')
        bad = np.flatnonzero(st & abi.ST_NAN)
        if bad.size:
            raise ValueErr(msg=f'At least one value is NaN! (rollout {int(bad[0])} of the batch, {bad.size} in total)')

    def slice(self, start: int, count: int):
        b = RolloutBatch(self.plan, self._out, self._dp, self._dp_index, self._env_spec, self._extra,
                         self._offset + start, count)
        b._host = self._host
        return b

    def __getitem__(self, i):
        if isinstance(i, slice):
            start, stop, step = i.indices(self._count)
            if step != 1:
                return [self[j] for j in range(start, stop, step)]
            return self.slice(start, max(0, stop - start))
        i = int(i)
        if i < 0:
            i += self._count
        if not 0 <= i < self._count:
            raise IndexError(i)
        g = self._offset + i
        L = int(self._h('length')[g])
        info = dict(env_spec=self._env_spec, domain_param=self.domain_param(g))
        if 'obs' not in self._out or L == 0:
            return RolloutSummary(self._h('ret')[g], L, info)
        done = np.zeros(L, dtype=bool)
        done[-1] = True
        obs, act, rew = self._trajectory(g, L)
        return StepSequence(observations=obs, actions=act, rewards=rew,
                            env_infos=dict(t=np.arange(L)*self.dt), rollout_info=info, done=done, complete=True)


class SummaryRollouts(Sequence):
    """ Rollouts known only by their results (returns, lengths, domain parameters): what a rank holds of the rollouts
    that ran on other GPUs.  Items are RolloutSummary objects built on access. """

    def __init__(self, ret: np.ndarray, length: np.ndarray, dp_of=None, env_spec=None):
        self._ret, self._length, self._dp_of, self._env_spec = np.asarray(ret), np.asarray(length), dp_of, env_spec

    def __len__(self):
        return int(self._ret.shape[0])

    def undiscounted_returns(self) -> np.ndarray:
        return self._ret

    @property
    def lengths(self) -> np.ndarray:
        return self._length

    @property
    def num_steps(self) -> int:
        return int(self._length.sum())

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        i = int(i)
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        info = dict(env_spec=self._env_spec, domain_param=None if self._dp_of is None else self._dp_of(i))
        return RolloutSummary(self._ret[i], self._length[i], info)


class GatheredBatch(Sequence):
    """ A sampling call under torch.distributed as every rank sees it: ALL n_global rollouts, in global order.  Returns,
    lengths, status words and (when each rollout drew its own) domain parameters were all-gathered; trajectories stay
    on the GPU that produced them, so rollouts [start, stop) of this rank are full items of its RolloutBatch and the
    others are RolloutSummary objects.  len(), undiscounted_returns(), lengths, num_steps and slicing therefore agree on
    every rank, as the reference's algorithms expect of the list sample() returns. """

    def __init__(self, local: RolloutBatch, gathered: dict, start: int, n_global: int):
        self.local, self.start, self.stop, self._n = local, start, start + len(local), n_global
        self.gathered = gathered
        self._host = {k: v.detach().cpu().numpy() for k, v in gathered.items()}

    def __len__(self):
        return self._n

    @property
    def device_results(self) -> dict:
        return self.local.device_results

    def undiscounted_returns(self) -> np.ndarray:
        return self._host['ret']

    @property
    def lengths(self) -> np.ndarray:
        return self._host['length']

    @property
    def num_steps(self) -> int:
        return int(self._host['length'].sum())

    def domain_param(self, i: int) -> dict:
        if self.start <= i < self.stop:
            return self.local.domain_param(i - self.start)
        if 'domain_param' not in self._host:
            return None
        return self.local.plan.dp_dicts(self._host['domain_param'], [i], self.local._extra)[0]

    def check(self):
        st = self._host['status']
        if np.any(self._host['length'] < 0) or np.any(st & abi.ST_INTERNAL):
            raise RuntimeError('the rollout kernel of some rank stored no result for a rollout or reported an internal fault')
        bad = np.flatnonzero(st & (abi.ST_BAD_INDEX | abi.ST_NAN))
        if bad.size:
            raise ValueErr(msg=f'At least one value is NaN or an index is outside its table! (rollout {int(bad[0])} of '
                               f'the batch, {bad.size} in total)')

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(self._n))]
        i = int(i)
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        if self.start <= i < self.stop:
            return self.local[i - self.start]
        info = dict(env_spec=self.local._env_spec, domain_param=self.domain_param(i))
        return RolloutSummary(self._host['ret'][i], self._host['length'][i], info)
