#!/usr/bin/env python3
""" For every closed-loop case of tools/stress_parity.py whose deviation from the C oracle exceeds 1e-9, show WHERE the
deviation comes from instead of asserting it:
  first   the first step at which the worst environment deviates by more than 1e-12, which quantity (obs / act / rew)
          crosses first, and the deviation there;
  growth  geometric growth per step from there to the end of the episode, next to the growth of the ORACLE's own
          deviation when its initial state is moved by one ulp (same environment);
  flip    first step at which a discontinuous branch is taken differently (QBallBalancer dead zone = recorded action
          exactly zero on one side only; action clip; the +-pi fold of the reward), or 'none';
  replay  the oracle's recorded actions through psh_step for the whole case (identical states, parameters, actions):
          the device dynamics alone;
  policy  the device policy alone on the oracle's states of the worst environment (one-step rollouts).
Usage: python tools/diagnose_outliers.py [seeds] [n_envs] """
import copy
import dataclasses
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simurlacra_amd import _abi as abi  # noqa: E402
from simurlacra_amd.engine import get_engine  # noqa: E402
from oracle import oracle_lib as orc  # noqa: E402
from tests import cases  # noqa: E402

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 25
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
eng = get_engine(0)
T = 50
KEYS = ('obs', 'act', 'rew')


def per_env_step_err(a, b):
    """ [T(+1), N] relative deviation per step and env, max over the component axis; NaN (past the end) -> 0 """
    out = {}
    for k in KEYS:
        x, y = a[k], b[k]
        if x.ndim == 2:
            x, y = x[:, None, :], y[:, None, :]
        e = np.abs(x - y)/np.maximum(1., np.abs(y))
        out[k] = np.nan_to_num(e, nan=0.).max(axis=1)
    return out


def describe(err, env, length):
    L = int(length[env])
    series = np.maximum(np.maximum(err['obs'][1:L + 1, env], err['act'][:L, env]), err['rew'][:L, env])
    over = np.flatnonzero(series > 1e-12)
    if over.size == 0:
        return None
    k0 = int(over[0])
    first = min(KEYS, key=lambda k: next((j for j in range(L) if err[k][j + (k == 'obs'), env] > 1e-12), 10**9))
    e0, e1 = float(series[k0]), float(series[L - 1])
    rate = (e1/e0)**(1./max(1, L - 1 - k0)) if e0 > 0 and e1 > 0 else float('nan')
    return dict(k0=k0, first=first, e0=e0, e1=e1, rate=rate, L=L)


def flips(kind, out, ref, env, length, case):
    """ first step where a branch of the step differs between device and oracle for this env """
    L = int(length[env])
    a, b = out['act'][:L, :, env], ref['act'][:L, :, env]
    found = []
    if kind == abi.ENV_QBB:
        z = np.flatnonzero(((a == 0.) != (b == 0.)).any(axis=1))
        if z.size:
            found.append(f'dead zone at step {int(z[0])}')
    alo, ahi = orc.constants(kind, case['dp'][:abi.ENV_DIMS[kind][3], env])[3:5]
    c = np.flatnonzero((((a > ahi) != (b > ahi)) | ((a < alo) != (b < alo))).any(axis=1))
    if c.size:
        found.append(f'action clip at step {int(c[0])}')
    return ', '.join(found) if found else 'none'


def replay(case, ref, flags=0):
    """ oracle's recorded actions through psh_step and the oracle's step: device dynamics alone; also returns the
    oracle's states per step """
    kind, dt = case['kind'], case['dt']
    S = abi.ENV_DIMS[kind][0]
    length = ref['length']
    steps = int(length.max())
    mem = eng.reset(kind, dt, T, case['task_c'], case['init'], case['dp'], flags=flags)
    mem_ref = orc.reset(kind, dt, T, case['task_c'], case['init'], case['dp'], flags=flags)
    acts = np.nan_to_num(ref['act'], nan=0.)
    states = [mem_ref['state'].copy()]
    worst = 0.
    for k in range(steps):
        o, r, d, st = eng.step(kind, dt, T, case['task_c'], mem, eng.f64(acts[k]), case['dp'], flags=flags)
        o_ref, r_ref, _, _, _ = orc.step(kind, dt, T, case['task_c'], mem_ref, acts[k], case['dp'], flags=flags)
        run = length > k
        states.append(mem_ref['state'].copy())
        worst = max(worst, cases.rel_err(o.cpu().numpy()[:, run], o_ref[:, run]), cases.rel_err(r.cpu().numpy()[run], r_ref[run]))
    return worst, states


def policy_alone(case, ref, env, states):
    """ device policy on the oracle's states of one env: one-step rollouts started from each of them """
    kind = case['kind']
    S = abi.ENV_DIMS[kind][0]
    L = int(ref['length'][env])
    init = np.stack([states[k][:S, env] for k in range(L)], axis=1)
    dp = np.tile(case['dp'][:, env:env + 1], (1, L))
    flags = abi.F_STOP_ON_DONE
    out = eng.rollout(kind, case['dt'], 1, case['task_c'], case['ps'], case['params'], init, dp, record=True, flags=flags)
    got = out['act'][0].cpu().numpy()                     # [A, L]
    return cases.rel_err(got, ref['act'][:L, :, env].T)


def report(label, kind, case, ref, out, args_for_sens=None, wrapped=False):
    length = ref['length']
    same = (out['length'] == length) & (out['status'] == ref['status'])
    err = per_env_step_err(out, ref)
    tot = np.maximum(np.maximum(err['obs'].max(axis=0), err['act'].max(axis=0)), err['rew'].max(axis=0))
    tot[~same] = np.inf
    env = int(np.argmax(tot))
    d = describe(err, env, length)
    line = f'{label}: worst env {env} err {tot[env]:.2e}, {int((tot > 1e-9).sum())}/{len(tot)} envs above 1e-9'
    if d:
        line += (f'\n    first   step {d["k0"]} of {d["L"]} ({d["first"]} first), deviation there {d["e0"]:.1e}'
                 f'\n    growth  x{d["rate"]:.3f} per step -> {d["e1"]:.1e} at the end')
    # oracle's own response to a one-ulp change of this env's initial state
    if args_for_sens is not None:
        ref2 = args_for_sens(np.nextafter(case['init'], np.inf))
        err2 = per_env_step_err(ref2, ref)
        d2 = describe(err2, env, np.minimum(length, ref2['length']))
        if d2:
            line += (f'; oracle vs itself (+1 ulp): first step {d2["k0"]}, {d2["e0"]:.1e}, x{d2["rate"]:.3f} per step -> '
                     f'{d2["e1"]:.1e}')
        else:
            line += '; oracle vs itself (+1 ulp): below 1e-12 throughout'
    line += f'\n    flip    {flips(kind, out, ref, env, length, case)}'
    if not wrapped:
        worst, states = replay(case, ref)
        line += f'\n    replay  device dynamics on the oracle\'s actions, all envs and steps: {worst:.1e}'
        line += f'\n    policy  device policy on the oracle\'s states of env {env}: {policy_alone(case, ref, env, states):.1e}'
    print(line, flush=True)


bad = 0
for kind in cases.KINDS:
    for pol, kw in (('linear', dict(scale=0.5)), ('fnn', dict(hidden=(64, 64), scale=0.5)), ('fnn', dict(hidden=(20, 7, 33), scale=0.5))):
        for seed in range(seeds):
            case = cases.make_case(kind, n, seed=1000*seed + kind, policy=pol, **kw)
            args = (kind, case['dt'], T, case['task_c'], case['ps'], case['params'])
            ref = orc.rollout(*args, case['init'], case['dp'], record=True)
            out = {k: v.cpu().numpy() for k, v in eng.rollout(*args, case['init'], case['dp'], record=True).items()}
            same = (out['length'] == ref['length']) & (out['status'] == ref['status'])
            e = max(cases.rel_err(out[k][..., same], ref[k][..., same]) for k in KEYS)
            if e > 1e-9 or not same.all():
                bad += 1
                report(f'{cases.KIND_NAMES[kind]} {pol} {kw.get("hidden", "")} seed {seed}', kind, case, ref, out,
                       lambda init2: orc.rollout(*args, init2, case['dp'], record=True))
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
            case = dict(case, dp=dp)
            ps = case['ps']
            if seed % 2:
                ps = dataclasses.replace(ps, expl_kind='normal' if seed % 4 == 1 else 'uniform', expl_mean=[0.01]*A, expl_scale=[0.2]*A)
            args = (kind, case['dt'], T, case['task_c'], ps, case['params'])
            kwargs = dict(record=True, wrap=wrap, bernoulli_reset=0.02 if seed % 3 == 0 else 0.)
            ref = orc.rollout(*args, case['init'], dp, **kwargs)
            out = {k: v.cpu().numpy() for k, v in eng.rollout(*args, case['init'], dp, **kwargs).items()}
            same = (out['length'] == ref['length']) & (out['status'] == ref['status'])
            e = max(cases.rel_err(out[k][..., same], ref[k][..., same]) for k in KEYS)
            if e > 1e-9 or not same.all():
                bad += 1
                report(f'wrapped {cases.KIND_NAMES[kind]} {"-".join(wrap.ops)} {pol} seed {seed}', kind, case, ref, out,
                       lambda init2: orc.rollout(*args, init2, dp, **kwargs), wrapped=True)
print(f'{bad} cases above 1e-9')
