#!/usr/bin/env python3
""" What the fused wrappers cost a large launch: one policy on n environments, bare and behind each wrapper.
Usage: python tools/perf_wrapped.py [n_envs] [linear|fnn] """
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simurlacra_amd import _abi as abi  # noqa: E402
from simurlacra_amd.engine import get_engine  # noqa: E402
from simurlacra_amd.specs import WrapSpec  # noqa: E402
from tests import cases  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
policy = sys.argv[2] if len(sys.argv) > 2 else 'fnn'
eng = get_engine(0)
for kind, name, T in ((abi.ENV_QQ, 'QQ', 600), (abi.ENV_QBB, 'QBB', 500)):
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    kw = dict(hidden=(64, 64), scale=0.2) if policy == 'fnn' else dict(feats=['identity'], scale=0.0)
    case = cases.make_case(kind, n, seed=1, policy=policy, **kw)
    args = (kind, case['dt'], T, case['task_c'], case['ps'], torch.as_tensor(case['params'], device=eng.device),
            torch.as_tensor(case['init'], device=eng.device), torch.as_tensor(case['dp'], device=eng.device))
    for tag, wrap in (('bare', None),
                      ('ActNormWrapper', WrapSpec(ops=['act_norm'])),
                      ('ActDelayWrapper(5)', WrapSpec(ops=['act_delay'], act_delay=5)),
                      ('GaussianObsNoiseWrapper', WrapSpec(ops=['obs_noise'], noise_seed=3, obs_noise_std=[0.001]*O)),
                      ('GaussianActNoiseWrapper', WrapSpec(ops=['act_noise'], noise_seed=3, act_noise_std=[0.001]*A)),
                      ('ObsNormWrapper(ActNormWrapper)', WrapSpec(ops=['obs_norm', 'act_norm']))):
        kw2 = {} if wrap is None else dict(wrap=wrap)
        out = eng.rollout(*args, final_state=False, **kw2)
        out = eng.rollout(*args, final_state=False, out=out, **kw2)
        ms = eng.last_rollout_ms()
        steps = int(out['length'].sum().item())
        print(f'{name:4s} {policy:6s} n={n} {tag:32s} mean_len={steps/n:6.1f} kernel {ms:8.2f} ms {steps/ms/1e6:8.3f} G env-steps/s',
              flush=True)
