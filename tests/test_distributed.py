"""
Multi-process tests of the env-sharding helpers on CPU (gloo backend, world_size 2): contiguous shard ranges and
the all-gather of per-rollout results, with even and uneven shard sizes.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from simurlacra_amd.sampling import distributed as D


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 1000, 2**21):
        for ws in (1, 2, 3, 8):
            ranges = [D.shard_range(n, r, ws) for r in range(ws)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(ranges[:-1], ranges[1:]))
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        start, stop = D.shard_range(n_total, rank, world)
        # "returns" of the global rollouts start..stop: a function of the GLOBAL index only
        ret = torch.arange(start, stop, dtype=torch.float64)**2
        length = torch.arange(start, stop, dtype=torch.int32) % 7
        dp = torch.stack([torch.arange(start, stop, dtype=torch.float64), -torch.arange(start, stop, dtype=torch.float64)], dim=1)
        g_ret, g_len, g_dp = D.all_gather_rows(ret, n_total), D.all_gather_rows(length, n_total), D.all_gather_rows(dp, n_total)
        ok = (torch.equal(g_ret, torch.arange(n_total, dtype=torch.float64)**2)
              and torch.equal(g_len, torch.arange(n_total, dtype=torch.int32) % 7)
              and torch.equal(g_dp[:, 1], -torch.arange(n_total, dtype=torch.float64)))
        q.put((rank, bool(ok), D.world()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n_total', [10, 11])
def test_all_gather_rows_world_size_2(n_total):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res) and all(r[2][1] == 2 for r in res)


# ------------------------------------------------------------------ the samplers under torch.distributed (gloo, CPU)
class _StubEngine:
    """ Stands in for the GPU engine in processes without one: every result is a function of the GLOBAL rollout index
    (and of the policy parameters), so the test can tell whether each rank sees every rollout in its place. """
    device = torch.device('cpu')

    def f64(self, a, shape=None):
        t = torch.as_tensor(np.asarray(a, dtype=np.float64) if not torch.is_tensor(a) else a, dtype=torch.float64).contiguous()
        return t if shape is None else t.reshape(shape)

    def i32(self, a):
        return None if a is None else torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a, dtype=torch.int32).contiguous()

    def randomize(self, n_envs, distrs, nominal_rows, seed, env_offset=0):
        out = torch.tensor(nominal_rows, dtype=torch.float64).reshape(-1, 1).repeat(1, n_envs)
        out[0] = 9.81 + 1e-3*(env_offset + torch.arange(n_envs, dtype=torch.float64))   # g carries the global index
        return out

    def sample_init(self, env_kind, n_envs, domain_param, seed, env_offset=0, dp_index=None):
        from simurlacra_amd import _abi as abi
        return torch.zeros((abi.ENV_DIMS[env_kind][4], n_envs), dtype=torch.float64)

    def rollout(self, env_kind, dt, max_steps, task_c, policy_spec, params, init_state, domain_param, *, flags=1, dp_index=None,
                policy_index=None, policy_group=0, record=False, wrap=None, bernoulli_reset=0., final_state=True, out=None):
        from simurlacra_amd import _abi as abi
        S, O, A, P, _ = abi.ENV_DIMS[env_kind]
        n, T = init_state.shape[1], int(max_steps)
        g = wrap.env_offset + torch.arange(n, dtype=torch.float64)
        params = self.f64(params).reshape(-1, policy_spec.num_param)
        ret = g.clone()
        if policy_group > 0:
            ret = params.sum(dim=1).repeat_interleave(policy_group) + 1e-3*(g % policy_group)
        res = dict(ret=ret, length=(g.to(torch.int32) % T) + 1, status=torch.zeros(n, dtype=torch.int32),
                   final_state=torch.zeros((S, n), dtype=torch.float64))
        if record:
            res.update(obs=torch.zeros((T + 1, O, n), dtype=torch.float64), act=torch.zeros((T, A, n), dtype=torch.float64),
                       rew=torch.zeros((T, n), dtype=torch.float64))
            res['rew'][0] = ret                               # the rewards of a trajectory add up to its return
        return res

    def segment_mean(self, ret, group_size):
        return ret.reshape(-1, group_size).mean(dim=1)


def _sampler_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import simurlacra_amd.engine as engine
        stub = _StubEngine()
        engine.get_engine = lambda device=None: stub
        from simurlacra_amd.domain_randomization import get_default_randomizer
        from simurlacra_amd.environment_wrappers import DomainRandWrapperLive
        from simurlacra_amd.environments import QQubeSim
        from simurlacra_amd.policies import FeatureStack, LinearPolicy, identity_feat
        from simurlacra_amd.sampling import ParallelSampler, ParameterExplorationSampler
        from simurlacra_amd.sampling.batch import RolloutSummary
        T = 7
        env = QQubeSim(dt=0.01, max_steps=T)
        renv = DomainRandWrapperLive(env, get_default_randomizer(env))
        pol = LinearPolicy(renv.spec, FeatureStack([identity_feat]))
        # ParallelSampler: 11 rollouts over 2 ranks (6 + 5); every rank sees all 11 in global order
        ros = ParallelSampler(renv, pol, 1, min_rollouts=11, seed=1).sample()
        failed = []

        def chk(idx, cond):
            if not cond:
                failed.append(idx)
        chk(0, len(ros) == 11)
        chk(1, np.array_equal(ros.undiscounted_returns(), np.arange(11.)))
        chk(2, np.array_equal(ros.lengths, np.arange(11) % T + 1) and ros.num_steps == int((np.arange(11) % T + 1).sum()))
        start, stop = D.shard_range(11, rank, world)
        for i in range(11):
            ro = ros[i]
            chk(3, ro.undiscounted_return() == float(i) and len(ro) == i % T + 1)
            chk(4, isinstance(ro, RolloutSummary) == (not start <= i < stop))          # trajectories stay local
            chk(5, abs(ro.rollout_info['domain_param']['g'] - (9.81 + 1e-3*i)) < 1e-12)   # gathered domain parameters
        chk(6, [r.undiscounted_return() for r in ros[3:8]] == [3., 4., 5., 6., 7.])
        # min_steps: the top-up loop sees the global step count on every rank and stops on the same round
        ms = ParallelSampler(renv, pol, 1, min_steps=40, seed=1).sample()
        chk(7, ms.num_steps >= 40 and len(ms) == len(ms.undiscounted_returns()))
        n_ms = len(ms)
        # ParameterExplorationSampler: 5 parameter sets (3 + 2) x 3 rollouts
        pop, R = 5, 3
        sets = torch.arange(pop*pol.num_param, dtype=torch.float64).reshape(pop, pol.num_param)
        res = ParameterExplorationSampler(renv, pol, 1, R, seed=2).sample(sets)
        want = sets.sum(dim=1).numpy()[:, None] + 1e-3*np.arange(R)[None, :]
        chk(8, len(res) == pop and res.num_rollouts == pop*R)
        chk(9, np.allclose(res.mean_returns, want.mean(axis=1), rtol=1e-14))
        for j in range(pop):
            chk(10, len(res[j].rollouts) == R and res[j].num_rollouts == R)
            chk(11, np.allclose([ro.undiscounted_return() for ro in res[j].rollouts], want[j], rtol=1e-14))
            chk(12, abs(res[j].mean_undiscounted_return - want[j].mean()) < 1e-9)
            chk(13, all(len(ro) >= 1 for ro in res[j].rollouts))
        half = res[1:]                                        # PEPG-style slicing of the result
        chk(14, len(half) == pop - 1 and np.allclose(half.mean_returns, want[1:].mean(axis=1)))
        q.put((rank, failed, n_ms))
    finally:
        dist.destroy_process_group()


def test_samplers_return_the_same_rollout_lists_on_every_rank():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sampler_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] == [] for r in res), [r[:2] for r in res]   # numbers of the failed checks per rank
    assert res[0][2] == res[1][2]


# ------------------------------------------------------------------ bench.py --gpus N starts its own ranks
def _bench(*argv, env=None):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', 'MASTER_ADDR')}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(root, 'bench.py')] + list(argv), capture_output=True, text=True,
                          timeout=300, env=e)


def test_bench_gpus_2_launches_two_ranks_itself():
    """ `bench.py --gpus 2` without a launcher around it: two ranks (gloo, the stub engine whose results are functions of the
    global env index), one JSON line from rank 0 with n_gpus = rccl_ranks = 2 and the env-steps of BOTH ranks. """
    import json
    sets, R, K, T = 2, 100, 3, 600
    r = _bench('--gpus', '2', '--backend', 'gloo', '--stub-engine', '--steps', str(K), '--warmup', '1', '--sets', str(sets),
               '--rollouts', str(R))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['rccl_ranks'] == 2 and d['steps'] == K and d['scaling'] == 'weak'
    n = 2*sets*R                                                       # envs of both ranks, lengths g % T + 1
    total = K*sum(g % T + 1 for g in range(n))
    assert d['value']*d['ms_per_step']*1e-3*K == pytest.approx(total, rel=1e-9)
    assert d['roofline']['kernel_launches_timed'] == K
    assert d['roofline']['env_steps_per_launch'] == sum(g % T + 1 for g in range(n//2))   # rank 0's share
    assert 'stub' in d['data']


def test_bench_fails_loudly_when_the_ranks_do_not_match():
    # more GPUs asked for than visible (none here, or fewer than 64 anywhere)
    r = _bench('--gpus', '64', '--steps', '1', '--warmup', '0')
    assert r.returncode != 0 and 'GPU(s) are visible' in r.stderr
    # a launcher that started a different number of ranks than --gpus names
    r = _bench('--gpus', '4', '--backend', 'gloo', '--stub-engine', '--steps', '1', '--warmup', '0', '--sets', '1', '--rollouts', '10',
               env=dict(WORLD_SIZE='1', RANK='0', LOCAL_RANK='0'))
    assert r.returncode != 0 and 'WORLD_SIZE' in r.stderr
