#!/usr/bin/env python3
""" Kernel time of FNN rollouts over the batch size (QQube, dt = 1/500, T = 600, FNN 6-64-64-1 tanh, randomized
parameters): small launches are latency bound, so how a launch is laid out over the chip decides their time.
Usage: python tools/perf_batch.py [hidden, e.g. 64,64] """
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simurlacra_amd import _abi as abi, envdefs  # noqa: E402
from simurlacra_amd.engine import get_engine  # noqa: E402
from tests import cases  # noqa: E402

hidden = tuple(int(h) for h in sys.argv[1].split(',')) if len(sys.argv) > 1 else (64, 64)
eng = get_engine(0)
kind, T = abi.ENV_QQ, 600
S, O, A, P, I = abi.ENV_DIMS[kind]
task_c = envdefs.default_task_spec(kind).to_c(S, A)
ps, params = cases.fnn_policy(kind, np.random.default_rng(2), hidden=hidden, scale=0.25)
for n in (64, 256, 1000, 4096, 16384, 32768, 65536, 100000, 131072, 262144, 524288):
    rng = np.random.default_rng(n)
    dp = torch.as_tensor(cases.domain_params(kind, n, rng), device=eng.device)
    init = torch.as_tensor(cases.init_states(kind, dp.cpu().numpy(), rng), device=eng.device)
    best = 1e9
    for it in range(3):
        out = eng.rollout(kind, 1/500., T, task_c, ps, params, init, dp, final_state=False)
        torch.cuda.synchronize()
        best = min(best, eng.last_rollout_ms())
    steps = int(out['length'].sum(dtype=torch.int64).item())
    print(f'n={n:7d} mean_len={steps/n:6.1f} max_len={int(out["length"].max()):4d} kernel {best:8.3f} ms  {steps/best/1e6:8.3f} G env-steps/s', flush=True)
