#!/usr/bin/env python3
""" Parity sweep beyond the test suite: many seeds x every env kind x LinearPolicy / FNNPolicy, kernels against the C
oracle.  A deviation above the tolerance is reported together with the oracle's own response to one-ulp changes of
the initial states and of the domain parameters (closed loops with a high gain amplify rounding differences; a kernel
defect shows as a deviation far above that).  Usage: python tools/stress_parity.py [seeds] [n_envs] """
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simurlacra_amd import _abi as abi  # noqa: E402
from simurlacra_amd.engine import get_engine  # noqa: E402
from oracle import oracle_lib as orc  # noqa: E402
from tests import cases  # noqa: E402

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
eng = get_engine(0)
T = 50
worst = 0.
bad = 0
for kind in cases.KINDS:
    for pol, kw in (('linear', dict(scale=0.5)), ('fnn', dict(hidden=(64, 64), scale=0.5)), ('fnn', dict(hidden=(20, 7, 33), scale=0.5))):
        for seed in range(seeds):
            case = cases.make_case(kind, n, seed=1000*seed + kind, policy=pol, **kw)
            args = (kind, case['dt'], T, case['task_c'], case['ps'], case['params'])
            ref = orc.rollout(*args, case['init'], case['dp'], record=True)
            out = {k: v.cpu().numpy() for k, v in eng.rollout(*args, case['init'], case['dp'], record=True).items()}
            same = (out['length'] == ref['length']) & (out['status'] == ref['status'])
            err = max(cases.rel_err(out[k][..., same], ref[k][..., same]) for k in ('obs', 'act', 'rew'))
            worst = max(worst, err if same.all() else np.inf)
            if err > 1e-9 or not same.all():
                sens, s2 = 0., np.ones(n, dtype=bool)
                for init2, dp2 in ((np.nextafter(case['init'], np.inf), case['dp']), (case['init'], case['dp']*(1 + 2.**-52))):
                    ref2 = orc.rollout(*args, init2, dp2, record=True)
                    s2 = s2 & (ref2['length'] == ref['length'])
                    sens = max(sens, max(cases.rel_err(ref2[k][..., s2], ref[k][..., s2]) for k in ('obs', 'act', 'rew')))
                bad += 1
                print(f'{cases.KIND_NAMES[kind]:7s} {pol:6s} {kw.get("hidden", "")} seed {seed}: err {err:.2e}, '
                      f'{int((~same).sum())} envs differ in length/status; oracle self-sensitivity {sens:.2e}, '
                      f'{int((~s2).sum())} lengths', flush=True)
    print(f'{cases.KIND_NAMES[kind]:7s} done', flush=True)
# the fused wrapper chains, exploration noise and the stochastic reset
import copy  # noqa: E402
import dataclasses  # noqa: E402
for kind, wrap0 in cases.WRAP_CASES:
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    for pol in ('linear', 'fnn'):
        for seed in range(seeds):
            wrap = copy.deepcopy(wrap0)
            wrap.noise_seed, wrap.env_offset = 77 + seed, 1000*seed
            kw = dict(obs_dim=wrap.obs_dim(O), scale=0.3)
            if pol == 'fnn':
                kw['hidden'] = (32, 32)
            case = cases.make_case(kind, n, seed=500*seed + kind, policy=pol, **kw)
            dp = case['dp']
            if wrap.act_delay_row >= 0:
                dp = np.vstack([dp, np.random.default_rng(seed).integers(0, 6, n).astype(float)[None, :]])
            ps = case['ps']
            if seed % 2:
                ps = dataclasses.replace(ps, expl_kind='normal' if seed % 4 == 1 else 'uniform', expl_mean=[0.01]*A, expl_scale=[0.2]*A)
            args = (kind, case['dt'], T, case['task_c'], ps, case['params'], case['init'], dp)
            kwargs = dict(record=True, wrap=wrap, bernoulli_reset=0.02 if seed % 3 == 0 else 0.)
            ref = orc.rollout(*args, **kwargs)
            out = {k: v.cpu().numpy() for k, v in eng.rollout(*args, **kwargs).items()}
            same = (out['length'] == ref['length']) & (out['status'] == ref['status'])
            err = max(cases.rel_err(out[k][..., same], ref[k][..., same]) for k in ('obs', 'act', 'rew'))
            if err > 1e-9 or not same.all():
                bad += 1
                print(f'wrapped {cases.KIND_NAMES[kind]:7s} {"-".join(wrap.ops)} {pol} seed {seed}: err {err:.2e}, '
                      f'{int((~same).sum())} envs differ in length/status', flush=True)
    print(f'wrapped {cases.KIND_NAMES[kind]:7s} {"-".join(wrap0.ops)} done', flush=True)
print(f'{bad} cases above 1e-9')
