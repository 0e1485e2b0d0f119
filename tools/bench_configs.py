#!/usr/bin/env python3
"""
The other workload configurations of the measurement plan, next to the headline one bench.py times (C5): throughput of
the rollout kernel from its HIP-event duration, one line per configuration.
  C1  QQubeSim dt=1/500 T=600, default randomizer, LinearPolicy(identity) W~N(0,0.1^2), 1000 rollouts; the NumPy
      restatement (one env at a time, as the reference runs) and the C restatement (OpenMP) are timed beside it
  C2  OneMassOscillator / BallOnBeam dt=0.01 T=500, N=2^20, LinearPolicy(identity), returns only and with trajectories
  C3  QCartPoleSwingUpSim dt=1/500 T=1000, FNN 64x64 tanh, N=2^21 per GPU
  C4  QBallBalancerSim dt=1/100 T=500, default randomizer + randomized act_delay (0..10), CEM population of 4096
      LinearPolicy(identity) x 8 rollouts on shared (domain parameter, init state) pairs
Usage: python tools/bench_configs.py [--quick]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simurlacra_amd import _abi as abi, envdefs  # noqa: E402
from simurlacra_amd.domain_randomization import get_default_randomizer, UniformDomainParam  # noqa: E402
from simurlacra_amd.engine import get_engine  # noqa: E402
from simurlacra_amd.environments import (BallOnBeamSim, OneMassOscillatorSim, QBallBalancerSim, QCartPoleSwingUpSim,  # noqa: E402
                                         QQubeSim)
from simurlacra_amd.specs import PolicySpec, WrapSpec  # noqa: E402

quick = '--quick' in sys.argv
eng = get_engine(0)
dev = eng.device


def setup(env, n, seed, extra_rows=()):
    kind = env.kind
    names = list(envdefs.DP_NAMES[kind]) + list(extra_rows)
    rand = get_default_randomizer(env)
    if extra_rows:
        rand.add_domain_params(UniformDomainParam(name='act_delay', mean=5, halfspan=5, clip_lo=0, roundint=True))
    nominal = list(env.domain_param_column()) + [0.]*len(extra_rows)
    dp = eng.randomize(n, rand.device_table(envdefs.DP_NAMES[kind], extra_rows), nominal, seed=seed)
    init = eng.sample_init(kind, n, dp, seed=seed)
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    return kind, dp, init, env.task_spec().to_c(S, A), (S, O, A)


def timed(tag, kind, dt, T, task_c, ps, params, init, dp, **kw):
    out = eng.rollout(kind, dt, T, task_c, ps, params, init, dp, final_state=False, **kw)
    out = eng.rollout(kind, dt, T, task_c, ps, params, init, dp, final_state=False, **kw)
    ms = eng.last_rollout_ms()
    steps = int(out['length'].sum().item())
    print(f'{tag:58s} N={init.shape[1]:8d} mean_len={steps/init.shape[1]:6.1f} kernel {ms:9.3f} ms {steps/ms/1e6:8.3f} G env-steps/s',
          flush=True)
    return out, steps, ms


# ---- C1
env = QQubeSim(dt=1/500., max_steps=600)
kind, dp, init, task_c, (S, O, A) = setup(env, 1000, 0)
ps = PolicySpec('linear', O, A, feats=['identity'])
W = 0.1*torch.randn(ps.num_param, dtype=torch.float64, generator=torch.Generator().manual_seed(0))
_, steps, ms = timed('C1 QQube linear(identity), 1000 rollouts (GPU)', kind, 1/500., 600, task_c, ps, W.to(dev), init, dp)
from oracle import oracle_lib as orc, pysim_oracle_np as onp  # noqa: E402
t0 = time.perf_counter()
o = orc.rollout(kind, 1/500., 600, task_c, ps, W.numpy(), init.cpu().numpy(), dp.cpu().numpy(), n_threads=16)
el = time.perf_counter() - t0
print(f'{"C1 same, C restatement (OpenMP, <=16 threads)":58s} {int(o["length"].sum())/el/1e6:8.3f} M env-steps/s')
n_np = 4 if quick else 20
t0 = time.perf_counter()
tot = 0
keys = list(envdefs.DP_NAMES[kind])
for i in range(n_np):
    e = onp.NpEnv('qq', 1/500., 600)
    ro = onp.rollout_np(e, onp.LinearPolicyNp(['identity'], W.numpy().reshape(A, -1)), init[:, i].cpu().numpy(),
                        dict(zip(keys, dp[:, i].cpu().numpy())))
    tot += ro['length']
el = time.perf_counter() - t0
print(f'{"C1 same, NumPy restatement (1 process, reference style)":58s} {tot/el/1e3:8.3f} k env-steps/s ({n_np} rollouts)')
# the same through a pool of worker processes, as the reference's ParallelSampler runs it; a child process of its own
# (this one holds the GPU)
import subprocess  # noqa: E402
workers = 8 if quick else 16
r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'np_pool_baseline.py'), str(workers),
                    str(workers*(2 if quick else 10))], capture_output=True, text=True, timeout=600)
print(f'{"C1 same, NumPy restatement in a process pool":58s} {r.stdout.strip() or r.stderr.strip()[-200:]}', flush=True)

# ---- C2
n = 1 << (16 if quick else 20)
for env, name in ((OneMassOscillatorSim(dt=0.01, max_steps=500), 'OMO'), (BallOnBeamSim(dt=0.01, max_steps=500), 'BOB')):
    kind, dp, init, task_c, (S, O, A) = setup(env, n, 1)
    ps = PolicySpec('linear', O, A, feats=['identity'])
    W = (0.1*torch.randn(ps.num_param, dtype=torch.float64, generator=torch.Generator().manual_seed(0))).to(dev)
    timed(f'C2 {name} linear(identity), returns only', kind, 0.01, 500, task_c, ps, W, init, dp)
    m = 65536
    out, steps, ms = timed(f'C2 {name} linear(identity), trajectories (first 65536 envs)', kind, 0.01, 500, task_c, ps, W,
                           init[:, :m].contiguous(), dp[:, :m].contiguous(), record=True)
    print(f'   trajectory writeback {(O + A + 1)*8*steps/ms/1e6:7.2f} GB/s of algorithmic bytes')

# ---- C3
n = 1 << (17 if quick else 21)
env = QCartPoleSwingUpSim(dt=1/500., max_steps=1000)
kind, dp, init, task_c, (S, O, A) = setup(env, n, 2)
ps = PolicySpec('fnn', O, A, hidden=[64, 64], hidden_nonlin=['tanh', 'tanh'])
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import reference_fnn_params  # noqa: E402
timed('C3 QCartPoleSwingUp FNN 64x64 tanh', kind, 1/500., 1000, task_c, ps, reference_fnn_params([O, 64, 64, A], 0).to(dev), init, dp)

# ---- C4
pop, R = (256 if quick else 4096), 8
env = QBallBalancerSim(dt=0.01, max_steps=500)
kind, dp_r, init_r, task_c, (S, O, A) = setup(env, R, 3, extra_rows=['act_delay'])
ps = PolicySpec('linear', O, A, feats=['identity'])
theta0 = 0.1*torch.randn(ps.num_param, dtype=torch.float64, generator=torch.Generator().manual_seed(0))
theta = (theta0 + 0.5*torch.randn((pop, ps.num_param), dtype=torch.float64, generator=torch.Generator().manual_seed(1))).to(dev)
r_idx = torch.arange(R, dtype=torch.int32, device=dev).repeat(pop)
init = init_r[:, r_idx.long()].contiguous()
wrap = WrapSpec(ops=['act_delay'], act_delay_row=abi.ENV_DIMS[kind][3])
out, steps, ms = timed(f'C4 QBB CEM pop={pop} x R={R}, linear(identity), act_delay 0..10', kind, 0.01, 500, task_c, ps, theta,
                       init, dp_r, dp_index=r_idx, policy_group=R, wrap=wrap)
means = eng.segment_mean(out['ret'], R)
print(f'   per-policy mean returns on the device: best {float(means.max()):.3f}, median {float(means.median()):.3f}')
