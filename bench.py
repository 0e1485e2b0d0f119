#!/usr/bin/env python3
"""
Headline benchmark: domain-randomized env-steps/sec (QQube, FNN policy) on N MI355X.

Workload (per GPU, weak scaling): the QQube parameter sweep (8 domain-parameter sets per GPU) — `--sets` domain-parameter sets
(randomizer distributions whose Mp/Mr/Lp/Lr means and stds are perturbed within +-30 % of the default randomizer)
x `--rollouts` rollouts each, QQubeSim dt = 1/500, T = 600, FNNPolicy 6-64-64-1 tanh with the reference's
initialisation (orthogonal weights, N(0,1) biases, seed 0).  One "step" = one pass of the hot path over that batch:
on-device domain randomization + init-state sampling + the fused rollout kernel + per-set return statistics
(+ one all-gather of those statistics across ranks).  Only env-steps executed before each env's `done` are counted.

Prints ONE JSON line on rank 0.

`--gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): this process starts N ranks itself, one per GPU
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, before anything here touches a GPU), relays rank 0's
JSON line and fails if fewer than N GPUs are visible or any rank fails.  Under `python -m torch.distributed.run` the
ranks are the launcher's; the line then reports the world size the collectives actually ran with (`rccl_ranks`), and a
`--gpus` that disagrees with it is an error.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from simurlacra_amd import _abi as abi, envdefs  # noqa: E402
from simurlacra_amd.engine import get_engine  # noqa: E402
from simurlacra_amd.specs import PolicySpec  # noqa: E402

FLOP_PER_STEP = 9352          # algorithmic fp64 flop per env-step, QQube + FNN 64x64 (transcendentals counted as 1)
FP64_PEAK_TFLOPS = 78.6       # MI355X dense FP64 (matrix = vector rate: half the 157.3 TF FP32 peak)
# HBM bytes per launch of the rollout kernel cannot be measured from inside this script: they come from the rocprofv3
# counter passes over this same command (tools/profile_bench.sh), whose summary is committed next to the kernel trace.
TRAFFIC_SUMMARIES = ['r03_bench_summary.json', 'r02_bench_summary.json', 'r01_bench_summary.json']   # newest first, under profiles/
DT, T = 1/500., 600


def reference_fnn_params(sizes, seed):
    """ FNNPolicy parameters initialised as Pyrado/pyrado/policies/initialization.py:26-40 does (orthogonal weights,
    standard-normal biases), flattened in parameters() order. """
    g = torch.Generator().manual_seed(seed)
    parts = []
    for fan_in, fan_out in zip(sizes[:-1], sizes[1:]):
        a = torch.randn((fan_out, fan_in), dtype=torch.float64, generator=g)
        # torch.nn.init.orthogonal_: QR of the (tall) Gaussian matrix with the sign fix on the diagonal of R
        tall = a if fan_out >= fan_in else a.t()
        q, r = torch.linalg.qr(tall)
        q = q*torch.sign(torch.diagonal(r)).unsqueeze(0)
        w = q if fan_out >= fan_in else q.t()
        parts += [w.reshape(-1), torch.randn(fan_out, dtype=torch.float64, generator=g)]
    return torch.cat(parts)


def qq_set_distrs(set_id):
    """ Randomizer of domain-parameter set `set_id`: the default QQube randomizer
    (Pyrado/pyrado/domain_randomization/default_randomizers.py:253-263) with the means and stds of Mp, Mr, Lp, Lr
    (the BayRn map, :413-429) scaled by factors in [0.7, 1.3] drawn from a generator seeded with the set id. """
    nom = envdefs.nominal_domain_param(abi.ENV_QQ)
    names = list(nom.keys())
    clip = dict(g=1e-3, Rm=1e-3, km=1e-4, Mr=1e-4, Lr=1e-4, Dr=1e-9, Mp=1e-4, Lp=1e-4, Dp=1e-9)
    rng = np.random.default_rng(1000 + set_id)
    distrs = []
    for name, lo in clip.items():
        mean, std = nom[name], nom[name]/5
        if name in ('Mp', 'Mr', 'Lp', 'Lr'):
            mean *= rng.uniform(0.7, 1.3)
            std *= rng.uniform(0.7, 1.3)
        distrs.append(dict(kind=abi.DISTR_NORMAL, row=names.index(name), a=mean, b=std, clip_lo=lo))
    return distrs, list(nom.values())


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)   # (the first two launches of a process run 2-3 % slower)
    ap.add_argument('--sets', type=int, default=8, help='domain-parameter sets per GPU')
    ap.add_argument('--rollouts', type=int, default=262144, help='rollouts per domain-parameter set')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help="torch.distributed backend of the ranks; 'gloo' with --stub-engine rehearses the multi-rank "
                         'plumbing on a machine without GPUs (tests/test_distributed.py)')
    ap.add_argument('--stub-engine', action='store_true', help=argparse.SUPPRESS)
    return ap.parse_args(argv)


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def launch_ranks(args, argv):
    """ One child process per GPU, started before this process has touched a GPU.  Rank 0's stdout is this run's stdout. """
    import subprocess
    n = args.gpus
    if not args.stub_engine:
        have = torch.cuda.device_count()     # (counting devices does not initialise the GPU)
        if have < n:
            sys.exit(f'bench.py: --gpus {n} asked for, but only {have} GPU(s) are visible to this process')
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv,
                                      env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = procs[0].communicate()[0].decode()
    codes = [p.wait() for p in procs]
    if any(codes):
        sys.stdout.write(out0)
        sys.exit(f'bench.py: ranks exited with {codes}')
    lines = [ln for ln in out0.splitlines() if ln.startswith('{')]
    if len(lines) != 1 or json.loads(lines[0]).get('n_gpus') != n or json.loads(lines[0]).get('rccl_ranks') != n:
        sys.stdout.write(out0)
        sys.exit(f'bench.py: expected one JSON line from rank 0 reporting {n} ranks')
    print(lines[0], flush=True)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus < 1:
        sys.exit('bench.py: --gpus must be >= 1')
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return launch_ranks(args, argv)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if '--gpus' in ' '.join(argv) and args.gpus != world:
        sys.exit(f'bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE)')
    gpu = not args.stub_engine
    sync = torch.cuda.synchronize if gpu else (lambda: None)
    # under torch.distributed.run the collectives run even with a single rank, so one GPU exercises the RCCL path
    if world > 1 or ('RANK' in os.environ and 'MASTER_PORT' in os.environ):
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if gpu and torch.cuda.device_count() <= local_rank:
            sys.exit(f'bench.py: rank {rank} has no GPU {local_rank} ({torch.cuda.device_count()} visible)')
        if gpu:
            torch.cuda.set_device(local_rank)
        # RCCL writes a version banner to stdout when its communicator is created: send it to stderr, stdout carries
        # the one JSON line only
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            if gpu and args.backend == 'nccl':
                dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
            else:
                dist.init_process_group(args.backend)
            dist.barrier()
            sync()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    else:
        dist = None
    if gpu:
        torch.cuda.set_device(local_rank)
        eng = get_engine(local_rank)
    else:
        eng = StubEngine()
    dev = eng.device

    S, O, A, P, _ = abi.ENV_DIMS[abi.ENV_QQ]
    task_c = envdefs.default_task_spec(abi.ENV_QQ).to_c(S, A)
    ps = PolicySpec('fnn', O, A, hidden=[64, 64], hidden_nonlin=['tanh', 'tanh'])
    params = reference_fnn_params([O, 64, 64, A], seed=0).to(dev)
    n_sets, R = args.sets, args.rollouts
    N = n_sets*R
    sets = [qq_set_distrs(rank*n_sets + j) for j in range(n_sets)]
    dp = torch.empty((P, N), dtype=torch.float64, device=dev)
    out = None

    def one_step(it):
        nonlocal out
        for j, (distrs, nominal) in enumerate(sets):
            g0 = (rank*n_sets + j)*R                       # global index of the set's first env
            dp[:, j*R:(j + 1)*R] = eng.randomize(R, distrs, nominal, seed=it, env_offset=g0)
        init = eng.sample_init(abi.ENV_QQ, N, dp, seed=it, env_offset=rank*N)
        out = eng.rollout(abi.ENV_QQ, DT, T, task_c, ps, params, init, dp, final_state=False, out=out)
        ret = out['ret'].view(n_sets, R)
        stats = torch.stack([ret.mean(dim=1), ret.var(dim=1), torch.full((n_sets,), float(R), device=dev,
                                                                         dtype=torch.float64)], dim=1)
        steps = out['length'].sum(dtype=torch.int64)
        if dist is not None:
            gathered = torch.empty((world*n_sets, 3), dtype=torch.float64, device=dev)
            dist.all_gather_into_tensor(gathered, stats)
            stats = gathered
        return steps, stats

    for w in range(args.warmup):
        one_step(-1 - w)
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    total_steps = torch.zeros((), dtype=torch.int64, device=dev)
    for it in range(args.steps):
        steps, stats = one_step(it)
        total_steps += steps
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0

    # whole-job numbers: max time over ranks, sum of executed env-steps over ranks, ranks that took part in the collectives
    tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    ranks = torch.ones((), dtype=torch.int64, device=dev)
    local_steps = int(total_steps.item())
    if dist is not None:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(total_steps, op=dist.ReduceOp.SUM)
        dist.all_reduce(ranks, op=dist.ReduceOp.SUM)
    elapsed = float(tt.item())
    total = int(total_steps.item())
    rccl_ranks = int(ranks.item())
    if rccl_ranks != world:
        sys.exit(f'bench.py: {rccl_ranks} ranks took part in the collectives, {world} were started')

    # dominant kernel: the rollout launches of the K timed steps themselves (rank 0's), each bracketed by HIP events on
    # the launch stream inside the library; read back only now, so the timed region ran without a host wait in it
    k_all = [eng.rollout_ms(back) for back in range(min(args.steps, 64))]
    if os.environ.get("PSH_BENCH_VERBOSE"): print("kernel ms, newest first:", [round(v, 2) for v in k_all], file=sys.stderr)
    assert all(v > 0. for v in k_all), k_all
    k_ms = float(np.mean(k_all))
    steps_k = local_steps/args.steps                      # env-steps of one launch, mean over the same launches
    achieved = FLOP_PER_STEP*steps_k/(k_ms*1e-3)/1e12

    executed = executed_fp64()
    if executed['executed_flop_per_env_step']:
        executed['executed_fp64_frac'] = executed['executed_flop_per_env_step']*steps_k/(k_ms*1e-3)/1e12/FP64_PEAK_TFLOPS

    traffic = traffic_source = None
    if (n_sets, R) == (8, 262144):  # the committed counter runs hold for the default workload only
        for name in TRAFFIC_SUMMARIES:
            path = os.path.join(ROOT, 'profiles', name)
            if os.path.exists(path):
                with open(path) as f:
                    traffic = json.load(f).get('hbm_bytes_per_launch')
                traffic_source = f'profiles/{name} (rocprofv3 FETCH_SIZE/WRITE_SIZE passes over this command, not this run)'
                break

    res = None
    if rank == 0:
        res = {
            'metric': 'domain-randomized env-steps/sec (QQube, FNN policy)',
            'value': total/elapsed, 'unit': 'env-steps/s', 'n_gpus': world, 'rccl_ranks': rccl_ranks if dist is not None else 0,
            'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3*elapsed/args.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic' if gpu else 'stub engine on CPUs: a rehearsal of the multi-rank plumbing, no measurement',
            'config': {'workload': 'QQubeSim dt=1/500 T=600, DomainRandomizer (default QQ randomizer, per-set Mp/Mr/Lp/Lr '
                                   'mean/std +-30%), FNNPolicy 6-64-64-1 tanh (orthogonal init seed 0)',
                       'domain_param_sets_per_gpu': n_sets, 'rollouts_per_set': R, 'envs_per_gpu': N,
                       'env_steps_counted': 'executed before done', 'parallelism': f'env-sharded x{world}'},
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': achieved/FP64_PEAK_TFLOPS, 'traffic': traffic, 'traffic_source': traffic_source,
                         'kernel': 'k_rollout_fnn<QQ,64>', 'kernel_ms': k_ms, 'kernel_ms_min_max': [min(k_all), max(k_all)],
                         'kernel_launches_timed': len(k_all), 'env_steps_per_launch': steps_k,
                         'flop_per_env_step': FLOP_PER_STEP, **executed},
        }
        if not args.no_cpu_baseline and world == 1 and gpu:
            res['cpu_baseline'] = cpu_baseline(eng, sets, task_c, ps, params)
            try:
                res['cpu_baseline']['slices'] = cpu_strong_slices(eng)
            except Exception as exc:  # a courtesy leg, as the pool below: its failure must not lose the GPU numbers
                res['cpu_baseline']['slices'] = {'error': repr(exc)}
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


class StubEngine:
    """ Stands in for the GPU engine when the multi-rank plumbing of this script is rehearsed on CPUs (--stub-engine with
    --backend gloo): results are functions of the global env index, kernel time is a constant.  Measures nothing. """
    device = torch.device('cpu')

    def randomize(self, n_envs, distrs, nominal_rows, seed, env_offset=0):
        out = torch.tensor(nominal_rows, dtype=torch.float64).reshape(-1, 1).repeat(1, n_envs)
        out[0] = env_offset + torch.arange(n_envs, dtype=torch.float64)
        return out

    def sample_init(self, env_kind, n_envs, domain_param, seed, env_offset=0, dp_index=None):
        return torch.zeros((abi.ENV_DIMS[env_kind][4], n_envs), dtype=torch.float64)

    def rollout(self, env_kind, dt, max_steps, task_c, policy_spec, params, init_state, domain_param, *, final_state=True, out=None):
        g = domain_param[0]
        return dict(ret=g.clone(), length=(g.to(torch.int32) % int(max_steps)) + 1, status=torch.zeros(g.numel(), dtype=torch.int32))

    def rollout_ms(self, back=0):
        return 1.0


def executed_fp64():
    """ Hardware fp64 flop the kernel executes per env-step, from the committed instruction-counter profile of this
    command (tools/pmc_insts.sh): MFMAs at 2 x 16 x 16 x 4 flop (the few 4x4x4 ones of the output layer and of draining waves
    counted alike: an overestimate of under 2 %), vector FMA at 2 and MUL / ADD at 1 flop per lane.  (The round-2 counter
    run launched the kernel twice per pass; its per-wave-step column is halved here.)  Next to the algorithmic `frac` it says
    how much of the pipe the launch keeps busy with arithmetic of any kind. """
    for name, launches in (('r03_bench_instruction_mix.txt', 1.), ('r02_bench_instruction_mix.txt', 2.)):
        path = os.path.join(ROOT, 'profiles', name)
        if not os.path.exists(path):
            continue
        per_ws = {}
        with open(path) as f:
            for line in f:
                w = line.split()
                if len(w) >= 5 and w[0].startswith('SQ_') and w[2:4] == ['per', 'wave-step']:
                    per_ws[w[0]] = float(w[4])/launches
        try:
            flop_ws = (per_ws['SQ_INSTS_MFMA']*2*16*16*4 + 64*(2*per_ws['SQ_INSTS_VALU_FMA_F64'] + per_ws['SQ_INSTS_VALU_MUL_F64']
                                                               + per_ws['SQ_INSTS_VALU_ADD_F64']))
        except KeyError:
            continue
        return {'executed_flop_per_env_step': flop_ws/64., 'executed_fp64_source': f'profiles/{name} (counter passes over this command, not this run)'}
    return {'executed_flop_per_env_step': None, 'executed_fp64_source': None}


def cpu_strong_slices(eng, n=10_000):
    """ 10 000-environment slices of the other workloads of the measurement plan -- OneMassOscillator and BallOnBeam with
    a LinearPolicy on identity features (T = 500), QCartPoleSwingUp with the FNN 64x64 (T = 1 000) -- on the C restatement
    with OpenMP over all host cores and, on the same inputs, through the rollout kernel; returns must agree. """
    from oracle import oracle_lib as orc
    from simurlacra_amd.domain_randomization import get_default_randomizer
    from simurlacra_amd.environments import BallOnBeamSim, OneMassOscillatorSim, QCartPoleSwingUpSim
    threads = orc.lib().orc_num_threads()
    res = []
    for tag, env, T, pol in (('OneMassOscillatorSim dt=0.01 T=500, LinearPolicy(identity)', OneMassOscillatorSim(dt=0.01, max_steps=500), 500, 'linear'),
                             ('BallOnBeamSim dt=0.01 T=500, LinearPolicy(identity)', BallOnBeamSim(dt=0.01, max_steps=500), 500, 'linear'),
                             ('QCartPoleSwingUpSim dt=1/500 T=1000, FNNPolicy 64x64 tanh', QCartPoleSwingUpSim(dt=1/500., max_steps=1000), 1000, 'fnn')):
        kind = env.kind
        S, O, A, P, _ = abi.ENV_DIMS[kind]
        rand = get_default_randomizer(env)
        dp = eng.randomize(n, rand.device_table(envdefs.DP_NAMES[kind], ()), list(env.domain_param_column()), seed=7)
        init = eng.sample_init(kind, n, dp, seed=7)
        task_c = env.task_spec().to_c(S, A)
        if pol == 'fnn':
            ps = PolicySpec('fnn', O, A, hidden=[64, 64], hidden_nonlin=['tanh', 'tanh'])
            w = reference_fnn_params([O, 64, 64, A], seed=0)
        else:
            ps = PolicySpec('linear', O, A, feats=['identity'])
            w = 0.1*torch.randn(ps.num_param, dtype=torch.float64, generator=torch.Generator().manual_seed(0))
        for _ in range(2):
            out = eng.rollout(kind, env.dt, T, task_c, ps, w.to(eng.device), init, dp, final_state=False)
        ms = eng.rollout_ms(0)
        g_ret, g_len = out['ret'].cpu().numpy(), out['length'].cpu().numpy()
        t0 = time.perf_counter()
        o = orc.rollout(kind, env.dt, T, task_c, ps, w.numpy(), init.cpu().numpy(), dp.cpu().numpy(), n_threads=threads)
        el = time.perf_counter() - t0
        res.append({'workload': tag, 'envs': n, 'env_steps': int(o['length'].sum()), 'cpu_value': float(o['length'].sum()/el),
                    'cpu_seconds': el, 'cores': threads, 'gpu_value': float(g_len.sum()/(ms*1e-3)), 'gpu_kernel_ms': ms,
                    'unit': 'env-steps/s', 'lengths_equal': bool(np.array_equal(g_len, o['length'])),
                    'return_mean_std_cpu': [float(o['ret'].mean()), float(o['ret'].std())],
                    'return_mean_std_gpu': [float(g_ret.mean()), float(g_ret.std())]})
    return res


def host_cpu():
    """ what the box reports about its host CPU: model string, logical CPUs of the whole machine, CPUs this process may
    run on """
    model = None
    try:
        with open('/proc/cpuinfo') as f:
            model = next((line.split(':', 1)[1].strip() for line in f if line.startswith('model name')), None)
    except OSError:
        pass
    try:
        allowed = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed = None
    return dict(cpu_model=model, host_cpu_count=os.cpu_count(), cpus_allowed=allowed)


def cpu_baseline(eng, sets, task_c, ps, params, target_s=12.0):
    """ The C oracle (scalar fp64 restatement of the reference, OpenMP over envs) on a bounded sample of the same
    workload, timed on this box's host cores with the OpenMP default (OMP_NUM_THREADS, the share of the machine a
    one-GPU job gets), next to the reference-style leg (one Python environment per rollout through a process pool). """
    from oracle import oracle_lib as orc
    threads = orc.lib().orc_num_threads()
    distrs, nominal = sets[0]

    def sample(n, seed):
        dp = eng.randomize(n, distrs, nominal, seed=seed, env_offset=0)
        init = eng.sample_init(abi.ENV_QQ, n, dp, seed=seed, env_offset=0)
        return dp.cpu().numpy(), init.cpu().numpy()

    p = params.cpu().numpy()
    dp, init = sample(4096, 12345)
    t0 = time.perf_counter()
    o = orc.rollout(abi.ENV_QQ, DT, T, task_c, ps, p, init, dp, n_threads=threads)
    probe = time.perf_counter() - t0
    rate = o['length'].sum()/probe
    n = int(min(max(4096, target_s*rate/max(1., o['length'].mean())), 4_000_000))
    dp, init = sample(n, 54321)
    t0 = time.perf_counter()
    o = orc.rollout(abi.ENV_QQ, DT, T, task_c, ps, p, init, dp, n_threads=threads)
    el = time.perf_counter() - t0
    res = {'value': float(o['length'].sum()/el), 'unit': 'env-steps/s', 'cores': threads, 'kind': 'port',
           'sample': f'{n} envs of the same workload (set 0), {int(o["length"].sum())} env-steps in {el:.1f} s, '
                     f'C restatement of the reference (oracle/pysim_oracle.c, -O2, OpenMP over envs, {threads} threads)'}
    res.update(host_cpu())
    try:
        res['reference_style'] = reference_style_baseline(min(threads, 16))
    except Exception as exc:  # the pool is a courtesy leg: its failure must not lose the GPU numbers
        res['reference_style'] = {'error': repr(exc)}
    return res


def reference_style_baseline(workers, target_s=5.0):
    """ The smallest workload (1 000 QQube rollouts with a LinearPolicy) in the reference's cost model (Pyrado/pyrado/sampling/parallel_sampler.py:104-123,
    sampler_pool.py:286-418): rollouts spread over a pool of worker processes, each stepping a NumPy environment in the
    interpreter.  Runs as a child process of its own (tools/np_pool_baseline.py): its pool forks, which a process
    holding a GPU context must not do. """
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'np_pool_baseline.py'), str(workers), '--seconds',
                          str(target_s), '--json'], capture_output=True, text=True, timeout=120, check=True).stdout
    return json.loads(out.strip().splitlines()[-1])


if __name__ == '__main__':
    main()
