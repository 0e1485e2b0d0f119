#!/usr/bin/env python3
""" Throughput of population-mode rollouts (ParameterExplorationSampler layout: pop parameter sets x R rollouts on shared
(domain parameter, init state) pairs) for FNN policies of several sizes.  Usage: python tools/perf_population.py """
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simurlacra_amd import _abi as abi  # noqa: E402
from simurlacra_amd.engine import get_engine  # noqa: E402
from tests import cases  # noqa: E402

eng = get_engine(0)
for kind, name, T in ((abi.ENV_QBB, 'QBB', 500), (abi.ENV_QQ, 'QQ', 600)):
    for hidden in ((16, 16), (32, 32), (64, 64)) if len(sys.argv) < 2 else (tuple(int(h) for h in sys.argv[1].split(",")),):
        for pop, R in ((128, 8), (256, 16), (1024, 8), (4096, 4), (4096, 8), (4096, 16), (512, 64), (65536, 16)):
            case = cases.make_case(kind, pop*R, seed=1, policy='fnn', hidden=hidden, scale=0.2)
            params = torch.as_tensor(0.2*np.random.default_rng(0).standard_normal((pop, case['ps'].num_param)), device=eng.device)
            args = (kind, case['dt'], T, case['task_c'], case['ps'], params, torch.as_tensor(case['init'], device=eng.device),
                    torch.as_tensor(case['dp'], device=eng.device))
            out = eng.rollout(*args, policy_group=R, final_state=False)
            out = eng.rollout(*args, policy_group=R, final_state=False, out=out)
            ms = eng.last_rollout_ms()
            steps = int(out['length'].sum().item())
            print(f'{name:4s} fnn{hidden} pop={pop} R={R:3d} mean_len={steps/(pop*R):6.1f} kernel {ms:8.2f} ms '
                  f'{steps/ms/1e6:7.3f} G env-steps/s', flush=True)
