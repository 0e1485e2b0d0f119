#!/usr/bin/env python3
""" Rollout-kernel throughput for every env kind, LinearPolicy and FNNPolicy 64x64 (returns only, T = 600 or the env's
default horizon): env-steps/s from the kernel's own HIP-event duration.
Usage: python tools/perf_envs.py [n_envs] [linear|fnn|''] [weight scale] [features]; scale 0 keeps most envs alive for all T steps. """
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simurlacra_amd import _abi as abi, envdefs  # noqa: E402
from simurlacra_amd.engine import get_engine  # noqa: E402
from tests import cases  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
only = sys.argv[2] if len(sys.argv) > 2 else ''
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
feats = tuple(sys.argv[4].split(',')) if len(sys.argv) > 4 else ('const', 'identity', 'squared')   # LinearPolicy feature stack
eng = get_engine(0)
names = {abi.ENV_QQ: 'QQ', abi.ENV_QCP_SU: 'QCP_SU', abi.ENV_QCP_ST: 'QCP_ST', abi.ENV_QBB: 'QBB', abi.ENV_BOB: 'BOB',
         abi.ENV_OMO: 'OMO', abi.ENV_PEND: 'PEND', abi.ENV_QQ_ST: 'QQ_ST'}
for kind in cases.KINDS:
    for pol in ('linear', 'fnn'):
        if only and only != pol:
            continue
        case = cases.make_case(kind, n, seed=kind, policy=pol, scale=scale, **(dict(hidden=(64, 64)) if pol == 'fnn' else dict(feats=feats)))
        T = 600
        args = (kind, case['dt'], T, case['task_c'], case['ps'], torch.as_tensor(case['params'], device=eng.device),
                torch.as_tensor(case['init'], device=eng.device), torch.as_tensor(case['dp'], device=eng.device))
        out = eng.rollout(*args, final_state=False)
        out = eng.rollout(*args, final_state=False, out=out)
        ms = eng.last_rollout_ms()
        steps = int(out['length'].sum().item())
        print(f'{names[kind]:7s} {pol:6s} n={n} mean_len={steps/n:7.1f} kernel {ms:8.2f} ms  {steps/ms/1e6:8.3f} G env-steps/s',
              flush=True)
