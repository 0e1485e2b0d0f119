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
from ..environment_wrappers.action_normalization import ActNormWrapper
from ..environment_wrappers.base import EnvWrapper
from ..environment_wrappers.domain_randomization import (DomainRandWrapper, DomainRandWrapperBuffer,
                                                          DomainRandWrapperLive, MetaDomainRandWrapper)
from ..environment_wrappers.utils import all_envs, inner_env
from ..environments.pysim.base import SimPyEnv
from ..utils.exceptions import ValueErr
from .step_sequence import StepSequence


class EnvPlan:
    """ Static analysis of an environment chain: the simulation at its core and the wrappers around it. """

    def __init__(self, env):
        self.env = env
        self.sim = inner_env(env)
        if not isinstance(self.sim, SimPyEnv):
            raise NotImplementedError(f'{type(self.sim).__name__} is not one of the simulations the rollout kernels '
                                      f'implement (QQube, QCartPole, QBallBalancer, BallOnBeam, OneMassOscillator)')
        self.kind = self.sim.kind
        self.S, self.O, self.A, self.P, self.I = abi.ENV_DIMS[self.kind]
        self.dr_wrapper = None       # outermost DomainRandWrapperLive/Buffer (Meta wrappers forward to it)
        self.act_wrappers = []       # action wrappers from the outside in
        for e in all_envs(env):
            if isinstance(e, MetaDomainRandWrapper):
                continue
            if isinstance(e, DomainRandWrapper):
                if self.dr_wrapper is None:
                    self.dr_wrapper = e
            elif isinstance(e, (ActNormWrapper, ActDelayWrapper)):
                self.act_wrappers.append(e)
            elif isinstance(e, EnvWrapper):
                raise NotImplementedError(f'{type(e).__name__} is not supported by the batched rollout engine')
        self.names = DP_NAMES[self.kind]

    @property
    def fusable_env(self) -> bool:
        return not self.act_wrappers

    def flags(self, stop_on_done=True) -> int:
        return self.sim.engine_flags() | (abi.F_STOP_ON_DONE if stop_on_done else 0)

    def dp_dicts(self, table: np.ndarray, index=None, extra=None):
        """ Domain-parameter dicts (as rollout_info['domain_param'] holds them) for the columns of a table. """
        base = self.env.domain_param
        cols = range(table.shape[1]) if index is None else index
        out = []
        for c in cols:
            d = dict(base)
            d.update({k: table[r, c] for r, k in enumerate(self.names)})
            if extra:
                d.update(extra)
            out.append(d)
        return out


def sample_domain_params(plan: EnvPlan, eng, n: int, seed: int, env_offset: int = 0):
    """ Domain parameters for n environments as a device table [P, M] plus an index [n] (or None when M == n).
    Live wrapper: drawn on the device from the randomizer's distributions (host torch.distributions only for
    distributions without a device sampler); buffer wrapper: the buffer is the table and the ring index advances by
    n; no wrapper: the environment's current parameters for every lane. """
    w = plan.dr_wrapper
    nominal = plan.sim.domain_param_column()
    if w is None:
        return eng.f64(nominal.reshape(-1, 1)), torch.zeros(n, dtype=torch.int32, device=eng.device)
    if isinstance(w, DomainRandWrapperBuffer) and w.buffer is not None:
        buf = w.buffer if isinstance(w.buffer, list) else [w.buffer]
        table = np.stack([plan.sim.domain_param_column(d) for d in buf], axis=1)
        start = w.ring_idx or 0
        idx = (start + np.arange(n)) % len(buf)
        w._ring_idx = int((start + n) % len(buf))
        return eng.f64(table), eng.i32(idx)
    rand = w.randomizer
    try:
        distrs = rand.device_table(plan.names)
    except NotImplementedError:
        rand.randomize(n)
        lst = rand.get_params(-1, format='list', dtype='numpy')
        lst = [lst] if isinstance(lst, dict) else lst
        return eng.f64(np.stack([plan.sim.domain_param_column({k: float(v) for k, v in d.items() if k in plan.names})
                                 for d in lst], axis=1)), None
    return eng.randomize(n, distrs, list(nominal), seed=seed, env_offset=env_offset), None


def sample_init_states(plan: EnvPlan, eng, n: int, dp, dp_index, seed: int, env_offset: int = 0):
    return eng.sample_init(plan.kind, n, dp, seed=seed, env_offset=env_offset, dp_index=dp_index)


def run_batch(plan: EnvPlan, eng, policy, init_state, dp, dp_index=None, *, max_steps=None, stop_on_done=True,
              record=True, params=None, policy_index=None):
    """ N rollouts on the device.  Fused kernel when the policy and the env chain allow it, else a step loop with the
    policy evaluated by torch on the whole batch.  Returns the dict of device tensors of Engine.rollout. """
    T = plan.env.max_steps if max_steps is None else max_steps
    if T == inf:
        raise ValueErr(msg='batched rollouts need a finite max_steps')
    T = int(T)
    S, A = plan.S, plan.A
    task_c = plan.sim.task_spec().to_c(S, A)
    spec = policy.fused_spec() if hasattr(policy, 'fused_spec') else None
    if spec is not None and plan.fusable_env:
        if params is None:
            params = policy.param_values.detach().to(torch.float64)
        return eng.rollout(plan.kind, plan.sim.dt, T, task_c, spec, params, init_state, dp,
                           flags=plan.flags(stop_on_done), dp_index=dp_index, policy_index=policy_index, record=record)
    if params is not None:
        raise NotImplementedError('per-environment policy parameters need a fusable policy and environment chain')
    return _run_batch_stepwise(plan, eng, policy, init_state, dp, dp_index, T, stop_on_done, record, task_c)


def _run_batch_stepwise(plan, eng, policy, init_state, dp, dp_index, T, stop_on_done, record, task_c):
    """ rollout() loop (Pyrado/pyrado/sampling/rollout.py:133-254) for N envs at once with an arbitrary torch policy. """
    kind, dt, dev = plan.kind, plan.sim.dt, eng.device
    init_state, dp = eng.f64(init_state), eng.f64(dp)
    N, O, A = init_state.shape[1], plan.O, plan.A
    flags = plan.sim.engine_flags()
    mem = eng.reset(kind, dt, T, task_c, init_state, dp, flags=flags, dp_index=dp_index)
    out = dict(ret=torch.zeros(N, dtype=torch.float64, device=dev), length=torch.zeros(N, dtype=torch.int32, device=dev),
               status=torch.zeros(N, dtype=torch.int32, device=dev))
    if record:
        out['obs'] = torch.full((T + 1, O, N), float('nan'), dtype=torch.float64, device=dev)
        out['act'] = torch.full((T, A, N), float('nan'), dtype=torch.float64, device=dev)
        out['rew'] = torch.full((T, N), float('nan'), dtype=torch.float64, device=dev)
    live = torch.ones(N, dtype=torch.bool, device=dev)
    pol_dev = next((p.device for p in policy.parameters()), torch.device('cpu')) if hasattr(policy, 'parameters') else torch.device('cpu')
    # action wrappers, outermost first: ActNorm uses the inner env's per-lane bounds, ActDelay a per-lane queue
    delays, queue = None, None
    for w in plan.act_wrappers:
        if isinstance(w, ActDelayWrapper):
            delays = torch.full((N,), int(w.delay), dtype=torch.int64, device=dev)
            queue = torch.zeros((int(w.delay) + 1, A, N), dtype=torch.float64, device=dev)
    if hasattr(policy, 'reset'):
        policy.reset()
    k = 0
    obs = mem['obs'].clone()
    while bool(live.any()) and k < T:
        if record:
            out['obs'][k][:, live] = obs[:, live]
        with torch.no_grad():
            act = policy(obs.t().to(pol_dev)).to(dev, torch.float64).reshape(N, A).t().contiguous()
        nan = torch.isnan(act).any(dim=0) | torch.isnan(obs).any(dim=0)
        rec_act = act.clone()
        for w in plan.act_wrappers:
            if isinstance(w, ActNormWrapper):
                lo, hi = _act_bounds(plan, dp, dp_index, N)
                act = lo + (act + 1)*(hi - lo)/2
            elif isinstance(w, ActDelayWrapper) and queue is not None and queue.shape[0] > 1:
                queue[-1] = act
                act = queue[0].clone()
                queue[:-1] = queue[1:].clone()
        mem_prev = {key: v.clone() for key, v in mem.items()}
        o, r, d, st = eng.step(kind, dt, T, task_c, mem, act, dp, flags=flags, dp_index=dp_index)
        # lanes that were not live keep their memory
        for key in mem:
            sel = ~live
            if mem[key].dim() == 2:
                mem[key][:, sel] = mem_prev[key][:, sel]
            else:
                mem[key][sel] = mem_prev[key][sel]
        stepped = live & ~nan
        if record:
            out['act'][k][:, stepped] = (act if not plan.act_wrappers else rec_act)[:, stepped]
            out['rew'][k][stepped] = r[stepped]
        out['ret'] += torch.where(stepped, r, torch.zeros_like(r))
        out['length'] += stepped.to(torch.int32)
        out['status'] |= torch.where(stepped, st, torch.zeros_like(st)) | ((live & nan).to(torch.int32)*abi.ST_NAN)
        obs = torch.where(stepped.unsqueeze(0), o, obs)
        done = d.bool() & stepped
        live = stepped & ~(done & stop_on_done) & ~((st & abi.ST_NAN) != 0)
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
    if k == abi.ENV_QQ:
        hi = cols[9:10]
    elif k in (abi.ENV_QCP_SU, abi.ENV_QCP_ST):
        hi = torch.full((1, N), 8., dtype=torch.float64, device=cols.device)
    elif k == abi.ENV_QBB:
        hi = torch.full((2, N), 3., dtype=torch.float64, device=cols.device)
    elif k == abi.ENV_BOB:
        hi = cols[4:5]/2.*cols[0:1]*3.
    else:
        hi = cols[1:2]
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
    objects unless someone iterates over all of them. """

    def __init__(self, plan: EnvPlan, out: dict, dp, dp_index=None, env_spec=None, extra_dp=None, offset=0, count=None):
        self.plan, self._out = plan, out
        self._dp, self._dp_index = dp, dp_index
        self._host = None
        self._env_spec = env_spec if env_spec is not None else plan.env.spec
        self._extra = extra_dp
        self._offset = offset
        self._count = int(out['ret'].shape[0]) - offset if count is None else count
        self.dt = plan.sim.dt

    def _h(self):
        if self._host is None:
            self._host = {k: v.detach().cpu().numpy() for k, v in self._out.items()}
            self._host['dp'] = self._dp.detach().cpu().numpy()
            self._host['dp_index'] = None if self._dp_index is None else self._dp_index.detach().cpu().numpy()
        return self._host

    def __len__(self):
        return self._count

    @property
    def device_results(self) -> dict:
        return self._out

    def undiscounted_returns(self) -> np.ndarray:
        return self._h()['ret'][self._offset:self._offset + self._count]

    @property
    def lengths(self) -> np.ndarray:
        return self._h()['length'][self._offset:self._offset + self._count]

    @property
    def num_steps(self) -> int:
        return int(self.lengths.sum())

    def domain_param(self, i: int) -> dict:
        h = self._h()
        col = i if h['dp_index'] is None else int(h['dp_index'][i])
        return self.plan.dp_dicts(h['dp'], [col], self._extra)[0]

    def check(self):
        """ Raise as rollout() does when an observation, action or state became NaN in any lane. """
        st = self._h()['status'][self._offset:self._offset + self._count]
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
        h = self._h()
        g = self._offset + i
        L = int(h['length'][g])
        info = dict(env_spec=self._env_spec, domain_param=self.domain_param(g))
        if 'obs' not in h or L == 0:
            return RolloutSummary(h['ret'][g], L, info)
        done = np.zeros(L, dtype=bool)
        done[-1] = True
        return StepSequence(observations=h['obs'][:L + 1, :, g], actions=h['act'][:L, :, g], rewards=h['rew'][:L, g],
                            env_infos=dict(t=np.arange(L)*self.dt), rollout_info=info, done=done, complete=True)
