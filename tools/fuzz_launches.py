#!/usr/bin/env python3
""" Randomized launches against the C oracle: every trial draws a simulation, a batch size (odd sizes included), a horizon,
a policy (LinearPolicy with plain or structured features, FNNPolicy of random depth / width / nonlinearity), how policies and
domain parameters are assigned (one policy, an index per environment, a population; a column per environment or an index
into a few sets), the flags (stop on done, trajectory layout, simplified dynamics), recording on or off, a wrapper chain,
exploration noise, the stochastic reset, the task (default or custom), the entry point (psh_rollout or the
non-allocating psh_rollout_static), and now and then a NaN weight or a NaN in an
initial state.  The combinations reach kernel variants that the
fixed cases of the test suite do not pair up.  Lengths and status words must agree exactly, everything else to 1e-9; a
larger deviation is printed with the oracle's own response to one-ulp changes of its inputs.
FUZZ_INF=1 also puts infinities into initial states (the special cases of IEEE division in the dynamics, saturating
activations on infinite observations).  FUZZ_WILD=1 makes single domain parameters zero, negative, tiny or huge
(frictionless joints, massless bodies, singular mass matrices).
Usage: python tools/fuzz_launches.py [trials] [first seed]     (FUZZ_SHOW=1: print the environments that differ;
FUZZ_SEEDS=a,b,c: these trials only; FUZZ_SIZES / FUZZ_HORIZONS: other batch sizes and horizons) """
import copy
import dataclasses
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simurlacra_amd import _abi as abi  # noqa: E402
from simurlacra_amd.engine import get_engine  # noqa: E402
from simurlacra_amd.specs import PolicySpec  # noqa: E402
from oracle import oracle_lib as orc  # noqa: E402
from tests import cases  # noqa: E402

INF = bool(os.environ.get('FUZZ_INF'))
WILD = bool(os.environ.get('FUZZ_WILD'))
GENTLE = False   # main(gentle=True): small gains and short horizons only, so that no closed loop amplifies rounding to 1e-9
DRY = bool(os.environ.get('FUZZ_DRY'))   # no GPU: draw the cases and run the oracle only (checks this script)
TOL = 1e-9
SIZES = [1, 3, 16, 17, 63, 64, 65, 100, 257, 1000, 1025, 2049, 4100, 20000, 70001]
HORIZONS = [1, 2, 7, 30, 60, 120, 250]
if os.environ.get('FUZZ_SIZES'):      # e.g. FUZZ_SIZES=131072,300001 FUZZ_HORIZONS=250,600: few, large launches (queue, drain)
    SIZES = [int(x) for x in os.environ['FUZZ_SIZES'].split(',')]
if os.environ.get('FUZZ_HORIZONS'):
    HORIZONS = [int(x) for x in os.environ['FUZZ_HORIZONS'].split(',')]
HIDDEN = [(16, 16), (32, 32), (64, 64), (64,), (8,), (20, 7, 33), (64, 64, 64), (48, 64), (64, 17), (5, 5, 5, 5)]
CUSTOM = {}
for kind_, spec_, _ in cases.custom_tasks():
    CUSTOM.setdefault(kind_, []).append(spec_)
WRAPS = {}
for kind_, wrap_ in cases.WRAP_CASES:
    WRAPS.setdefault(kind_, []).append(wrap_)


def draw(rng):
    kind = cases.KINDS[rng.integers(len(cases.KINDS))]
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    t = dict(kind=kind, T=int(rng.choice(HORIZONS)), record=bool(rng.random() < (0.6 if max(SIZES) < 100000 else 0.)))
    wrap = None
    if kind in WRAPS and rng.random() < 0.4:
        wrap = copy.deepcopy(WRAPS[kind][rng.integers(len(WRAPS[kind]))])
        wrap.noise_seed, wrap.env_offset = int(rng.integers(1 << 30)), int(rng.integers(10000))
    t['wrap'] = wrap
    obs_dim = O if wrap is None else wrap.obs_dim(O)
    # how the policies are assigned
    mode = rng.choice(['one', 'one', 'index', 'population'])
    n = int(rng.choice(SIZES))
    n_pol, R = 1, 0
    if mode == 'index':
        n_pol = int(rng.integers(2, 10))
    elif mode == 'population':
        R = int(rng.choice([1, 3, 4, 8, 13, 16, 20, 64]))
        n_pol = max(1, n // R)
        n = n_pol*R
    t.update(mode=mode, n=n, n_pol=n_pol, R=R)
    # the policy
    scale = float(rng.choice([0.1, 0.3, 0.5, 1.0]))
    if GENTLE:
        scale, t['T'] = min(scale, 0.3), min(t['T'], 60)
    pk = rng.choice(['lin-id', 'lin-feats', 'lin-struct', 'fnn', 'fnn'])
    if pk == 'lin-id':
        ps = PolicySpec('linear', obs_dim, A, feats=[['identity'], ['const', 'identity']][rng.integers(2)])
    elif pk == 'lin-feats':
        names = ['const', 'identity', 'sign', 'abs', 'squared', 'cubed', 'sig', 'bell', 'sin', 'cos', 'sinsin']
        ps = PolicySpec('linear', obs_dim, A, feats=list(rng.choice(names, size=int(rng.integers(1, 6)), replace=False)))
    elif pk == 'lin-struct':
        ps = PolicySpec('linear', obs_dim, A, feats=cases.structured_feats(obs_dim, rng, state_wise=bool(rng.integers(2))))
    else:
        hidden = HIDDEN[rng.integers(len(HIDDEN))]
        nl = rng.choice(['tanh', 'tanh', 'tanh', 'relu', 'sigmoid'])
        nls = [nl]*len(hidden) if rng.random() < 0.8 else [str(rng.choice(['tanh', 'relu', 'sigmoid', 'none'])) for _ in hidden]
        ps = PolicySpec('fnn', obs_dim, A, hidden=list(hidden), hidden_nonlin=[str(x) for x in nls],
                        out_nonlin=[None, None, 'tanh', 'sigmoid'][rng.integers(4)])
    if wrap is not None and rng.random() < 0.3:
        ps = dataclasses.replace(ps, expl_kind=str(rng.choice(['normal', 'uniform'])), expl_mean=[0.01]*A, expl_scale=[0.2]*A)
    params = scale*rng.standard_normal((n_pol, ps.num_param))
    if rng.random() < 0.08:   # a NaN weight in one policy
        params[rng.integers(n_pol), rng.integers(ps.num_param)] = np.nan
    t.update(pk=str(pk), ps=ps, params=params)
    # inputs
    M = n if rng.random() < 0.6 else int(rng.integers(1, 9))
    dp = cases.domain_params(kind, M, rng)
    dp_index = None if M == n else rng.integers(0, M, n).astype(np.int32)
    dp_env = dp if dp_index is None else dp[:, dp_index]
    t['init'] = cases.init_states(kind, dp_env, rng)
    if rng.random() < 0.08:   # a few initial states with a NaN in them (FUZZ_INF=1: or an infinity, see the docstring)
        for _ in range(int(rng.integers(1, 4))):
            t['init'][rng.integers(t['init'].shape[0]), rng.integers(n)] = rng.choice([np.nan, np.inf, -np.inf] if INF else [np.nan])
    if WILD and rng.random() < 0.3:   # FUZZ_WILD=1: domain parameters that are zero, negated, tiny or huge in a few columns
        for _ in range(int(rng.integers(1, 4))):
            r, c = rng.integers(dp.shape[0]), rng.integers(M)
            dp[r, c] = [0., -dp[r, c], 1e-12*dp[r, c], 1e12*dp[r, c]][rng.integers(4)]
    if wrap is not None and wrap.act_delay_row >= 0:
        dp = np.vstack([dp, rng.integers(0, 6, M).astype(float)[None, :]])
    t.update(dp=dp, dp_index=dp_index)
    t['policy_index'] = rng.integers(0, n_pol, n).astype(np.int32) if mode == 'index' else None
    # flags, task
    flags = abi.F_STOP_ON_DONE if rng.random() < 0.75 else 0
    if t['record'] and rng.random() < 0.4:
        flags |= abi.F_TRAJ_ENV_MAJOR
    if kind == abi.ENV_QBB and rng.random() < 0.3:
        flags |= abi.F_SIMPLIFIED_DYN
    t['flags'] = flags
    t['custom_task'] = bool(kind in CUSTOM and rng.random() < 0.3)
    spec = CUSTOM[kind][rng.integers(len(CUSTOM[kind]))] if t['custom_task'] else cases.envdefs.default_task_spec(kind)
    t['task_c'] = spec.to_c(S, A)
    t['bernoulli'] = 0.02 if rng.random() < 0.15 else 0.
    t['static'] = bool(rng.random() < 0.15)   # through psh_rollout_static (scratch in the library's static device array)
    return t


def describe(t):
    ps = t['ps']
    pol = f"{t['pk']}{tuple(ps.hidden) if ps.kind == 'fnn' else ''}{'/' + '-'.join(map(str, ps.hidden_nonlin)) if ps.kind == 'fnn' else ''}"
    return (f"{cases.KIND_NAMES[t['kind']]:6s} n={t['n']:5d} T={t['T']:3d} {pol} {t['mode']}(pol={t['n_pol']},R={t['R']}) "
            f"dp={'env' if t['dp_index'] is None else t['dp'].shape[1]} flags={t['flags']} rec={int(t['record'])} "
            f"wrap={'-' if t['wrap'] is None else '+'.join(t['wrap'].ops)} expl={ps.expl_kind} bern={t['bernoulli']} "
            f"task={'custom' if t['custom_task'] else 'default'}{' static' if t.get('static') else ''}")


def run(fn, t, init=None, dp=None, to_np=False):
    out = fn(t['kind'], cases.DT[t['kind']], t['T'], t['task_c'], t['ps'], t['params'], t['init'] if init is None else init,
             t['dp'] if dp is None else dp, flags=t['flags'], dp_index=t['dp_index'], policy_index=t['policy_index'],
             policy_group=t['R'], record=t['record'], wrap=t['wrap'], bernoulli_reset=t['bernoulli'],
             **(dict(static_scratch=True) if to_np and t.get('static') else {}))
    if to_np:
        out = {k: v.cpu().numpy() for k, v in out.items()}
        if t['record'] and t['flags'] & abi.F_TRAJ_ENV_MAJOR:   # [N, T(+1), .] -> the oracle's [T(+1), ., N]
            out['obs'], out['act'], out['rew'] = out['obs'].transpose(1, 2, 0), out['act'].transpose(1, 2, 0), out['rew'].T
    return out


def show(t, out, ref, keys, limit=4):
    """ The environments whose results differ, with their inputs. """
    n = t['n']
    diff = (out['length'] != ref['length']) | (out['status'] != ref['status'])
    for k in keys:
        a, b = out[k].reshape(-1, n), ref[k].reshape(-1, n)
        with np.errstate(invalid='ignore'):
            close = (np.abs(a - b) <= TOL*np.maximum(1., np.abs(b))) | (a == b) | (np.isnan(a) & np.isnan(b))
        diff = diff | ~close.all(axis=0)
    for i in np.flatnonzero(diff)[:limit]:
        print(f'    env {i}: init {t["init"][:, i]} length {out["length"][i]} / {ref["length"][i]} status {out["status"][i]} / {ref["status"][i]}')
        for k in keys:
            a, b = out[k].reshape(-1, n)[:, i], ref[k].reshape(-1, n)[:, i]
            with np.errstate(invalid='ignore'):
                bad_rows = np.flatnonzero(~((np.abs(a - b) <= TOL*np.maximum(1., np.abs(b))) | (a == b) | (np.isnan(a) & np.isnan(b))))
            if bad_rows.size:
                r = bad_rows[0]
                print(f'      {k}: {bad_rows.size} entries, first at flat row {r}: {a[r]!r} / {b[r]!r}')
        if not np.isfinite(t['params']).all():
            print(f'      (params hold a NaN in policy {np.flatnonzero(~np.isfinite(t["params"]).all(axis=1))})')


def worst_err(out, ref, same, keys):
    return max(cases.rel_err(out[k][..., same], ref[k][..., same]) for k in keys)


def main(seeds, eng, inf=None, wild=None, log=print, gentle=False):
    """ Runs the trials with these seeds on `eng` (None: the oracle against itself); returns the numbers of unexplained
    mismatches and of refused combinations. """
    global INF, WILD, GENTLE
    INF, WILD, GENTLE = (INF if inf is None else inf), (WILD if wild is None else wild), gentle
    print = log   # noqa: A001
    bad = unsupported = 0
    t_start = time.time()
    for trial, seed in enumerate(seeds):
        rng = np.random.default_rng(seed)
        t = draw(rng)
        oflags = t['flags'] & ~abi.F_TRAJ_ENV_MAJOR
        t_or = dict(t, flags=oflags)
        try:
            ref = run(orc.rollout, t_or)
        except Exception as ex:   # the oracle refuses the combination: not a case
            unsupported += 1
            print(f'[{seed}] oracle: {type(ex).__name__}: {ex} | {describe(t)}', flush=True)
            continue
        try:
            out = ref if eng is None else run(eng.rollout, t, to_np=True)
        except Exception as ex:
            unsupported += 1
            print(f'[{seed}] engine: {type(ex).__name__}: {str(ex)[:150]} | {describe(t)}', flush=True)
            continue
        keys = ['ret', 'final_state'] + (['obs', 'act', 'rew'] if t['record'] else [])
        same = (out['length'] == ref['length']) & (out['status'] == ref['status'])
        try:
            err = worst_err(out, ref, same, keys)
        except AssertionError as ex:
            err = np.inf
            print(f'[{seed}] {ex}', flush=True)
        if err > TOL or not same.all():
            sens, s2 = 0., np.ones(t['n'], dtype=bool)
            for init2, dp2 in ((np.nextafter(t['init'], np.inf), t['dp']), (t['init'], t['dp']*(1 + 2.**-52))):
                ref2 = run(orc.rollout, t_or, init=init2, dp=dp2)
                s2 = s2 & (ref2['length'] == ref['length']) & (ref2['status'] == ref['status'])
                try:
                    sens = max(sens, worst_err(ref2, ref, s2, keys))
                except AssertionError:
                    sens = np.inf
            # (lengths: an episode that ends on a state bound can end a step earlier or later once a deviation has grown to the
            # distance from that bound; where one-ulp input changes already move the oracle's own values by 1e-3 and more, a
            # handful of such episodes among tens of thousands is that growth, not a defect)
            n_len = int((~same).sum())
            explained = err <= 100*max(sens, 1e-12) and (n_len <= 2*int((~s2).sum()) or (sens >= 1e-3 and n_len <= 1 + t['n']//10000))
            bad += 0 if explained else 1
            if not explained and os.environ.get('FUZZ_SHOW'):
                show(t, out, ref, keys)
            print(f'[{seed}] {"(explained) " if explained else "MISMATCH "}err {err:.2e}, {int((~same).sum())} envs differ in length/status; '
                  f'oracle self-sensitivity {sens:.2e}, {int((~s2).sum())} lengths | {describe(t)}', flush=True)
        if (trial + 1) % 25 == 0:
            print(f'... {trial + 1} trials, {time.time() - t_start:.0f} s', flush=True)
    return bad, unsupported


if __name__ == '__main__':
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    seeds = [int(x) for x in os.environ['FUZZ_SEEDS'].split(',')] if os.environ.get('FUZZ_SEEDS') else range(seed0, seed0 + trials)
    bad, unsupported = main(seeds, None if DRY else get_engine(0))
    print(f'{len(seeds)} trials from seed {seeds[0]}: {bad} unexplained mismatches, {unsupported} combinations refused')
