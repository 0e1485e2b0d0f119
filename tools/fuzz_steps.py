#!/usr/bin/env python3
""" Randomized calls of the entry points next to the fused rollout, against the C oracle:
  * psh_reset + psh_step driven with the same random actions on both sides (inside and outside the action bounds, zeros
    for QBallBalancer's dead zone, now and then a NaN), random batch sizes, domain parameters per environment or through
    an index, default and custom tasks, simplified dynamics, the infinite-horizon flag, stepping on after `done`:
    observation, reward, done, status, step counter, state and the action array as the step leaves it, after every step;
  * psh_randomize with random lists of distributions (normal, uniform, Bernoulli, clipping, rounding);
  * psh_sample_init for every simulation;
  * psh_topk (ties, NaNs) and psh_segment_mean against NumPy.
Usage: python tools/fuzz_steps.py [trials] [first seed] """
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simurlacra_amd import _abi as abi  # noqa: E402
from simurlacra_amd.engine import get_engine  # noqa: E402
from oracle import oracle_lib as orc  # noqa: E402
from tests import cases  # noqa: E402

eng = None     # set by main()
log = print
TOL = 1e-9
CUSTOM = {}
for kind_, spec_, _ in cases.custom_tasks():
    CUSTOM.setdefault(kind_, []).append(spec_)
bad = 0


def report(seed, what, msg):
    global bad
    bad += 1
    log(f'[{seed}] {what}: {msg}')


def close(a, b, tol=TOL):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if not np.array_equal(np.isnan(a), np.isnan(b)):
        return False
    with np.errstate(invalid='ignore'):
        ok = (a == b) | (np.abs(a - b) <= tol*np.maximum(1., np.abs(b))) | np.isnan(b)
    return bool(ok.all())


def fuzz_steps(seed, rng):
    kind = cases.KINDS[rng.integers(len(cases.KINDS))]
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    n = int(rng.choice([1, 5, 64, 65, 300, 1000, 4097]))
    T = int(rng.choice([1, 3, 10, 25]))
    steps = T + int(rng.integers(0, 4))           # on past the time limit
    M = n if rng.random() < 0.6 else int(rng.integers(1, 6))
    dp = cases.domain_params(kind, M, rng)
    dp_index = None if M == n else rng.integers(0, M, n).astype(np.int32)
    init = cases.init_states(kind, dp if dp_index is None else dp[:, dp_index], rng)
    spec = CUSTOM[kind][rng.integers(len(CUSTOM[kind]))] if kind in CUSTOM and rng.random() < 0.3 else cases.envdefs.default_task_spec(kind)
    task_c = spec.to_c(S, A)
    flags = (abi.F_SIMPLIFIED_DYN if kind == abi.ENV_QBB and rng.random() < 0.3 else 0) | (abi.F_MAX_STEPS_INF if rng.random() < 0.2 else 0)
    what = f'steps {cases.KIND_NAMES[kind]} n={n} T={T} dp={"env" if dp_index is None else M} flags={flags}'
    kw = dict(flags=flags, dp_index=dp_index)
    m_ref = orc.reset(kind, cases.DT[kind], T, task_c, init, dp, **kw)
    m_dev = eng.reset(kind, cases.DT[kind], T, task_c, init, dp, **kw)
    if not close(m_dev['obs'].cpu().numpy(), m_ref['obs'], 1e-13) or not close(m_dev['state'].cpu().numpy(), m_ref['state'], 1e-13):
        return report(seed, what, 'reset differs')
    ahi = np.stack([orc.constants(kind, dp[:, j])[4] for j in range(M)], axis=1)    # [A, M]
    ahi = ahi if dp_index is None else ahi[:, dp_index]
    for k in range(steps):
        act = rng.standard_normal((A, n))*ahi*rng.choice([0.01, 0.3, 1.0, 3.0])
        act[:, rng.random(n) < 0.05] = 0.
        if rng.random() < 0.15:
            act[rng.integers(A), rng.integers(n)] = np.nan
        o, r, d, s, a_left = orc.step(kind, cases.DT[kind], T, task_c, m_ref, act, dp, **kw)
        a_dev = torch.as_tensor(act.copy(), device=eng.device)
        o2, r2, d2, s2 = eng.step(kind, cases.DT[kind], T, task_c, m_dev, a_dev, dp, **kw)
        got = dict(obs=o2.cpu().numpy(), rew=r2.cpu().numpy(), act=a_dev.cpu().numpy(), state=m_dev['state'].cpu().numpy())
        want = dict(obs=o, rew=r, act=a_left, state=m_ref['state'])
        for key in want:
            if not close(got[key], want[key]):
                with np.errstate(invalid='ignore'):
                    e = np.nanmax(np.abs(got[key] - want[key]))
                return report(seed, what, f'step {k}: {key} differs (max abs {e:.2e})')
        for key, g, w in (('done', d2.cpu().numpy(), d), ('status', s2.cpu().numpy(), s),
                          ('curr_step', m_dev['curr_step'].cpu().numpy(), m_ref['curr_step']),
                          ('flags_io', m_dev['flags_io'].cpu().numpy(), m_ref['flags_io'])):
            if not np.array_equal(g, w):
                return report(seed, what, f'step {k}: {key} differs in {int((g != w).sum())} envs')


def fuzz_randomize(seed, rng):
    rows = int(rng.integers(1, 21))
    nominal = list(rng.standard_normal(rows)*3)
    distrs = []
    for row in rng.permutation(rows)[:int(rng.integers(0, rows + 1))]:
        kind = int(rng.choice([abi.DISTR_NORMAL, abi.DISTR_UNIFORM, abi.DISTR_BERNOULLI]))
        q = dict(kind=kind, row=int(row), a=float(rng.normal()*3), b=float(abs(rng.normal()) + 0.01), c=float(rng.random()))
        if rng.random() < 0.3:
            q['clip_lo'] = float(q['a'] - rng.random())
        if rng.random() < 0.3:
            q['clip_up'] = float(q['a'] + rng.random())
        if rng.random() < 0.2:
            q['roundint'] = 1
        distrs.append(q)
    n = int(rng.choice([1, 63, 1000, 100001]))
    s, off = int(rng.integers(1 << 62)), int(rng.integers(1 << 40))
    ref = orc.randomize(n, distrs, nominal, seed=s, env_offset=off)
    got = eng.randomize(n, distrs, nominal, seed=s, env_offset=off).cpu().numpy()
    if not close(got, ref, 1e-13):
        report(seed, f'randomize rows={rows} n={n}', f'max abs {np.abs(got - ref).max():.2e}; {distrs}')


def fuzz_sample_init(seed, rng):
    kind = cases.KINDS[rng.integers(len(cases.KINDS))]
    n = int(rng.choice([1, 65, 5000]))
    M = n if rng.random() < 0.5 else int(rng.integers(1, 6))
    dp = cases.domain_params(kind, M, rng)
    dp_index = None if M == n else rng.integers(0, M, n).astype(np.int32)
    s, off = int(rng.integers(1 << 62)), int(rng.integers(1 << 40))
    ref = orc.sample_init(kind, n, dp, seed=s, env_offset=off, dp_index=dp_index)
    got = eng.sample_init(kind, n, dp, seed=s, env_offset=off, dp_index=dp_index).cpu().numpy()
    if not close(got, ref, 1e-13):
        report(seed, f'sample_init {cases.KIND_NAMES[kind]} n={n}', f'max abs {np.abs(got - ref).max():.2e}')


def fuzz_stats(seed, rng):
    n = int(rng.choice([1, 2, 50, 1000, 5000]))
    v = rng.standard_normal(n).round(int(rng.choice([1, 8])))   # rounding to one digit makes ties
    v[rng.random(n) < 0.02] = np.nan
    k = int(rng.integers(1, n + 1))
    idx, val = eng.topk(v, k)
    idx, val = idx.cpu().numpy(), val.cpu().numpy()
    key = np.where(np.isnan(v), np.inf, v)                      # NaN first, as torch.argsort(descending=True)
    want = np.sort(key)[::-1][:k]
    if not (np.array_equal(np.where(np.isnan(val), np.inf, val), want) and len(set(idx.tolist())) == k
            and np.array_equal(np.where(np.isnan(v[idx]), np.inf, v[idx]), want)):
        report(seed, f'topk n={n} k={k}', 'values or indices differ')
    g = int(rng.choice([1, 3, 8, 50]))
    r = rng.standard_normal(g*int(rng.integers(1, 400)))
    got = eng.segment_mean(r, g).cpu().numpy()
    if not close(got, r.reshape(-1, g).mean(axis=1), 1e-12):
        report(seed, f'segment_mean g={g}', 'differs')


def main(seeds, engine, out=print):
    """ Runs the trials with these seeds on `engine`; returns the number of mismatches. """
    global eng, log, bad
    eng, log, bad = engine, out, 0
    for trial, seed in enumerate(seeds):
        rng = np.random.default_rng(seed)
        fuzz_steps(seed, rng)
        fuzz_randomize(seed, rng)
        fuzz_sample_init(seed, rng)
        fuzz_stats(seed, rng)
        if (trial + 1) % 50 == 0:
            out(f'... {trial + 1} trials')
    return bad


if __name__ == '__main__':
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    n_bad = main(range(seed0, seed0 + trials), get_engine(0), out=lambda m: print(m, flush=True))
    print(f'{trials} trials from seed {seed0}: {n_bad} mismatches')
