#!/usr/bin/env python3
"""
Reference-style CPU leg of configuration C1: the NumPy restatement of the reference (one Python environment object
per rollout, stepped one step at a time) run through a pool of worker processes, the way the reference's
ParallelSampler spreads rollouts over `num_envs` workers (Pyrado/pyrado/sampling/parallel_sampler.py:104-123,
sampler_pool.py:286-418).  No GPU is touched by this process or its workers.
Usage: python tools/np_pool_baseline.py [workers] [rollouts]   ->   prints env-steps/s
"""
import multiprocessing as mp
import os
import sys
import time

# one BLAS thread per worker: the 4-vector products of a step gain nothing from more, and the workers are the parallelism
for _v in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):
    os.environ[_v] = '1'

import numpy as np  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pysim_oracle_np as onp  # noqa: E402

DT, T = 1/500., 600


def one_rollout(seed):
    rng = np.random.default_rng(seed)
    nom = onp.nominal('qq')
    # the default QQube randomizer: Normal(nominal, nominal/5) with a positive lower clip, max_volt left alone
    dp = {k: (v if k == 'max_volt' else max(1e-9, rng.normal(v, v/5))) for k, v in nom.items()}
    init = rng.uniform(-1, 1, 4)*np.array([2, 1, .5, .5])/180*np.pi
    W = 0.1*np.random.default_rng(0).standard_normal((1, 6))
    ro = onp.rollout_np(onp.NpEnv('qq', DT, T), onp.LinearPolicyNp(['identity'], W), init, dp)
    return ro['length']


if __name__ == '__main__':
    import argparse
    import json
    ap = argparse.ArgumentParser()
    ap.add_argument('workers', nargs='?', type=int, default=min(16, os.cpu_count()))
    ap.add_argument('rollouts', nargs='?', type=int, default=None, help='rollouts to time (default: from --seconds)')
    ap.add_argument('--seconds', type=float, default=5.0, help='target duration when no rollout count is given')
    ap.add_argument('--json', action='store_true')
    args = ap.parse_args()
    workers = args.workers
    with mp.Pool(workers) as pool:
        pool.map(one_rollout, range(workers))          # start the workers before the clock
        t0 = time.perf_counter()
        steps = sum(pool.map(one_rollout, range(100, 100 + workers), chunksize=1))
        n = workers
        if args.rollouts is None:
            more = workers*int(max(1, min(40, args.seconds/max(time.perf_counter() - t0, 1e-3) - 1)))
        else:
            more = max(0, args.rollouts - workers)
        if more:
            steps += sum(pool.map(one_rollout, range(1000, 1000 + more), chunksize=1))
            n += more
        el = time.perf_counter() - t0
    if args.json:
        print(json.dumps({'value': steps/el, 'unit': 'env-steps/s', 'cores': workers, 'kind': 'port',
                          'sample': f'{n} rollouts of QQubeSim (dt 1/500, T 600) + default randomizer + LinearPolicy(identity)'
                                    f', {steps} env-steps in {el:.1f} s, NumPy restatement stepped in the '
                                    f'interpreter by {workers} worker processes'}))
    else:
        print(f'{steps/el:.1f} env-steps/s with {workers} worker processes ({n} rollouts, {steps} steps, {el:.1f} s)')
