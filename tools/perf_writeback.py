#!/usr/bin/env python3
""" Trajectory writeback of the LinearPolicy rollout kernels at a batch several times the resident lane count:
OneMassOscillator and BallOnBeam, default randomizers, LinearPolicy(identity), T = 500, record=True.  Reports the kernel
duration (HIP events), the algorithmic trajectory bytes ((O + A + 1) doubles per executed step + the final observation)
and the resulting GB/s; run under `rocprofv3 --pmc WRITE_SIZE` for the bytes the memory system counted.
Usage: python tools/perf_writeback.py [n_envs] [omo|bob|...] [returns_only|env_major] """
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simurlacra_amd import _abi as abi, envdefs  # noqa: E402
from simurlacra_amd.engine import get_engine  # noqa: E402
from tests import cases  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 21
which = sys.argv[2].split(',') if len(sys.argv) > 2 and sys.argv[2] else ['omo', 'bob']
mode = sys.argv[3] if len(sys.argv) > 3 else 'time_major'
record = mode != 'returns_only'
env_major = mode == 'env_major'
eng = get_engine(0)
dev = eng.device
KINDS = dict(omo=abi.ENV_OMO, bob=abi.ENV_BOB, qq=abi.ENV_QQ, qbb=abi.ENV_QBB, qcp=abi.ENV_QCP_SU)
T = 500
for name in which:
    kind = KINDS[name]
    S, O, A, P, I = abi.ENV_DIMS[kind]
    nom = envdefs.nominal_domain_param(kind)
    names = list(nom.keys())
    distrs = [dict(kind=abi.DISTR_NORMAL, row=r, a=v, b=abs(v)/5, clip_lo=1e-3*abs(v)) for r, v in enumerate(nom.values()) if v != 0]
    dp = eng.randomize(n, distrs, list(nom.values()), seed=1)
    init = eng.sample_init(kind, n, dp, seed=1)
    ps = cases.PolicySpec('linear', O, A, feats=['identity'])
    params = torch.as_tensor(0.1*np.random.default_rng(0).standard_normal(ps.num_param), device=dev)
    task_c = envdefs.default_task_spec(kind).to_c(S, A)
    out = dict(ret=torch.empty(n, dtype=torch.float64, device=dev), length=torch.empty(n, dtype=torch.int32, device=dev),
               status=torch.empty(n, dtype=torch.int32, device=dev))
    if record:   # left uninitialised: rows past an episode's end are never read here
        shapes = ((n, T + 1, O), (n, T, A), (n, T)) if env_major else ((T + 1, O, n), (T, A, n), (T, n))
        out.update(obs=torch.empty(shapes[0], dtype=torch.float64, device=dev),
                   act=torch.empty(shapes[1], dtype=torch.float64, device=dev), rew=torch.empty(shapes[2], dtype=torch.float64, device=dev))
    for it in range(2):
        eng.rollout(kind, cases.DT[kind], T, task_c, ps, params, init, dp, final_state=False, out=out,
                    flags=abi.F_STOP_ON_DONE | (abi.F_TRAJ_ENV_MAJOR if env_major else 0))
        ms = eng.last_rollout_ms()
    steps = int(out['length'].sum(dtype=torch.int64).item())
    traj = (steps*(O + A + 1) + n*O)*8 if record else 0
    fixed = n*((I + P)*8 + 16)
    print(f'{name:4s} n={n} T={T} record={record} layout={mode} mean_len={steps/n:6.1f} kernel {ms:8.2f} ms  {steps/ms/1e6:8.3f} G env-steps/s  '
          f'trajectory {traj/1e9:7.2f} GB -> {traj/ms/1e6:7.1f} GB/s  (+ {fixed/1e9:.2f} GB inputs and results)', flush=True)
    if record and not env_major:
        # time-major rows: the envs of a queue chunk start together, so a group of g neighbours keeps its g x 8-byte
        # segment of every row alive until its longest episode ends
        L = out['length'].cpu().numpy().astype(np.int64)
        pred = {g: float(L[:n//g*g].reshape(-1, g).max(axis=1).sum()*g)/float(L.sum()) for g in (4, 8, 16)}
        print('     bytes of whole segments / algorithmic bytes, were the memory system to write segments of '
              + ', '.join(f'{8*g} B: {v:.2f}' for g, v in pred.items()), flush=True)
    del out
    torch.cuda.empty_cache()
