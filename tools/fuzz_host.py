#!/usr/bin/env python3
""" Randomized rollouts through the Pyrado-style host interface: rollout(env, policy) takes the fused kernel when the
environment's wrapper chain and the policy can be fused, and the reference's own loop -- env.reset() / policy(obs) /
env.step() over the wrapper OBJECTS, the policy evaluated by torch -- otherwise.  Every trial builds a random simulation,
a random chain of deterministic wrappers in random order (ObsNormWrapper, ActNormWrapper, ActDelayWrapper,
ObsPartialWrapper), a random LinearPolicy (any of the fusable feature functions, MultFeat / RandFourierFeat / RBFFeat)
or FNNPolicy (random sizes and torch nonlinearities), and runs the same rollout both ways: lengths equal, observations,
actions and rewards within 1e-9 (or within what a rounding error on the initial state does to the stepwise rollout
itself).
Every third trial also runs the chain behind DomainRandWrapperLive in a ParallelSampler, fused against the batched step
loop of a policy that cannot be fused.  Three implementations meet here: the kernels' features, networks and wrappers, torch's, and the host wrappers.
Usage: python tools/fuzz_host.py [trials] [first seed] """
import os
import sys

import numpy as np
import torch as to

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simurlacra_amd.environment_wrappers import ActDelayWrapper, ActNormWrapper, ObsNormWrapper, ObsPartialWrapper  # noqa: E402
from simurlacra_amd.environments import (BallOnBeamSim, OneMassOscillatorSim, PendulumSim, QBallBalancerSim,  # noqa: E402
                                         QCartPoleStabSim, QCartPoleSwingUpSim, QQubeSim, QQubeStabSim)
from simurlacra_amd.policies import FeatureStack, FNNPolicy, LinearPolicy, MultFeat, RandFourierFeat, RBFFeat  # noqa: E402
from simurlacra_amd.policies import features as F  # noqa: E402
from simurlacra_amd.sampling import rollout  # noqa: E402
from simurlacra_amd.sampling.rollout import _fused_plan, _rollout_stepwise  # noqa: E402
from simurlacra_amd.utils.data_types import RenderMode  # noqa: E402
from tests import cases  # noqa: E402

TOL = 1e-9
SIMS = [(QQubeSim, 0.004), (QCartPoleSwingUpSim, 0.002), (QCartPoleStabSim, 0.002), (QBallBalancerSim, 0.01), (BallOnBeamSim, 0.01),
        (OneMassOscillatorSim, 0.01), (PendulumSim, 0.02), (QQubeStabSim, 0.004)]
PLAIN = [F.const_feat, F.identity_feat, F.sign_feat, F.abs_feat, F.squared_feat, F.qubic_feat, F.sig_feat, F.bell_feat,
         F.sin_feat, F.cos_feat, F.sinsin_feat, F.sincos_feat]


def build(rng, stage):
    """ stage = the random decisions of this trial, drawn once and replayed for the second copy of the environment """
    cls, dt = SIMS[stage['sim']]
    env = cls(dt=dt, max_steps=stage['T'])
    names = []
    for w in stage['wrappers']:
        if w == 'obs_norm':
            sp = env.obs_space
            ub = {sp.labels[i]: 25. for i in np.flatnonzero(~np.isfinite(sp.bound_up))}
            lb = {sp.labels[i]: -25. for i in np.flatnonzero(~np.isfinite(sp.bound_lo))}
            env = ObsNormWrapper(env, explicit_lb=lb or None, explicit_ub=ub or None)
        elif w == 'act_norm':
            env = ActNormWrapper(env)
        elif w == 'act_delay':
            env = ActDelayWrapper(env, delay=stage['delay'])
        else:
            n = env.obs_space.flat_dim
            mask = stage['mask'][:n] + [False]*(n - len(stage['mask'][:n]))
            if all(mask):
                mask[0] = False
            env = ObsPartialWrapper(env, mask=mask)
        names.append(w)
    return env


def make_policy(rng, env, stage):
    spec = env.spec
    O = spec.obs_space.flat_dim
    if stage['policy'] == 'linear':
        feats = [PLAIN[i] for i in stage['feats']]
        if stage['struct'] and O >= 2:
            feats += [MultFeat([0, O - 1]), RandFourierFeat(O, 8, bandwidth=np.full(O, 2.)),
                      RBFFeat(5, (np.full(O, -1.5), np.full(O, 1.5)), state_wise_norm=stage['state_wise'])]
        pol = LinearPolicy(spec, FeatureStack(feats))
    else:
        nl = {'tanh': to.tanh, 'relu': to.relu, 'sigmoid': to.sigmoid}
        pol = FNNPolicy(spec, stage['hidden'], [nl[x] for x in stage['nls']], output_nonlin=nl.get(stage['out_nl']))
    gen = to.Generator().manual_seed(stage['pseed'])
    pol.param_values = stage['scale']*to.randn(pol.num_param, dtype=to.float64, generator=gen)
    return pol


def max_dev(a, b):
    worst = 0.
    for x, y in ((a.observations, b.observations), (a.actions, b.actions), (a.rewards, b.rewards)):
        worst = max(worst, cases.rel_err(np.asarray(x), np.asarray(y)))
    return worst


def main(seeds, log=print, gentle=False):
    """ Runs the trials with these seeds; returns the numbers of unexplained mismatches, fused rollouts, chains that could
    not be built and sampler pairs. """
    print = lambda *a, **k: log(' '.join(map(str, a)))   # noqa: A001, E731
    bad = fused = skipped = samplers = 0
    for trial, seed in enumerate(seeds):
        rng = np.random.default_rng(seed)
        hidden = [(16, 16), (32, 32), (64, 64), (20, 7), (8,), (64, 64, 64), (48, 64)][rng.integers(7)]
        stage = dict(sim=int(rng.integers(len(SIMS))), T=int(rng.choice([10, 40, 80])),
                     wrappers=list(rng.permutation(['obs_norm', 'act_norm', 'act_delay', 'obs_partial'])[:rng.integers(0, 5)]),
                     delay=int(rng.integers(0, 5)), mask=[bool(x) for x in rng.integers(0, 2, 8)],
                     policy=str(rng.choice(['linear', 'fnn'])), feats=[int(i) for i in rng.permutation(len(PLAIN))[:rng.integers(1, 5)]],
                     struct=bool(rng.random() < 0.3), state_wise=bool(rng.integers(2)), hidden=list(hidden),
                     nls=[str(rng.choice(['tanh', 'tanh', 'relu', 'sigmoid'])) for _ in hidden],
                     out_nl=str(rng.choice(['none', 'none', 'tanh'])), pseed=int(rng.integers(1 << 30)),
                     scale=float(rng.choice([0.1, 0.3, 1.0])), stop=bool(rng.random() < 0.8))
        if gentle:   # small gains and short horizons only: no closed loop that amplifies rounding to 1e-9
            stage['scale'], stage['T'] = min(stage['scale'], 0.3), min(stage['T'], 40)
        what = (f"{SIMS[stage['sim']][0].__name__} T={stage['T']} {'>'.join(stage['wrappers']) or '-'} delay={stage['delay']} "
                f"{stage['policy']} {[PLAIN[i].__name__ for i in stage['feats']] if stage['policy'] == 'linear' else (stage['hidden'], stage['nls'], stage['out_nl'])}"
                f"{' +struct' if stage['struct'] and stage['policy'] == 'linear' else ''} scale={stage['scale']} stop={stage['stop']}")
        try:
            env_a, env_b = build(rng, stage), build(rng, stage)
            to.manual_seed(stage['pseed'])
            np.random.seed(stage['pseed'] % (1 << 31))
            pol = make_policy(rng, env_a, stage)
        except Exception as ex:   # e.g. an ObsPartialWrapper mask that the chosen order makes invalid
            skipped += 1
            print(f'[{seed}] not built ({type(ex).__name__}: {str(ex)[:100]}) | {what}', flush=True)
            continue
        sim = env_a
        while hasattr(sim, 'wrapped_env'):
            sim = sim.wrapped_env
        dp = cases.domain_params(sim.kind, 1, rng)
        init = cases.init_states(sim.kind, dp, rng)[:, 0]
        dp_dict = dict(zip(sim.get_nominal_domain_param().keys(), dp[:, 0]))
        is_fused = _fused_plan(env_a, pol, bernoulli=False, no_reset=False, dts=False, render=False, recurrent=False) is not None
        fused += int(is_fused)

        def stepwise(env, init_state):
            return _rollout_stepwise(env, pol, True, dict(init_state=init_state.copy(), domain_param=dict(dp_dict)), RenderMode(), 1, None,
                                     False, False, False, stage['stop'])
        try:
            ro = rollout(env_a, pol, eval=True, reset_kwargs=dict(init_state=init.copy(), domain_param=dict(dp_dict)), stop_on_done=stage['stop'])
        except Exception as ex:
            try:
                stepwise(env_b, init)
                bad += 1
                print(f'[{seed}] MISMATCH rollout() raised {type(ex).__name__}: {str(ex)[:120]}, the stepwise loop did not | {what}', flush=True)
            except Exception as ex2:
                same = type(ex2) is type(ex)
                bad += 0 if same else 1
                print(f'[{seed}] {"both raise" if same else "MISMATCH raise"} {type(ex).__name__} / {type(ex2).__name__}: {str(ex)[:80]} | {what}', flush=True)
            continue
        try:
            ref = stepwise(env_b, init)
        except Exception as ex:
            bad += 1
            print(f'[{seed}] MISMATCH the stepwise loop raised {type(ex).__name__}: {str(ex)[:120]}, rollout() did not | {what}', flush=True)
            continue
        try:
            dev = np.inf if len(ro) != len(ref) else max_dev(ro, ref)
        except AssertionError:
            dev = np.inf
        if dev > TOL:
            # A closed loop with a high gain amplifies rounding differences.  Such a deviation is told from a defect by its
            # parts: the torch policy on the fused rollout's observations reproduces the fused actions (the policy alone), the
            # first steps agree to rounding (wrappers and dynamics alone), and the stepwise rollout itself moves when its
            # initial state moves by a rounding error.
            with to.no_grad():
                act_t = np.stack([np.atleast_1d(pol(to.as_tensor(o)).numpy()) for o in ro.observations[:-1]]) if len(ro) else np.zeros((0, 1))
            # (QBallBalancerSim writes its dead zone into the recorded action: those entries are exact zeros, not the policy's output)
            live = np.asarray(ro.actions) != 0.
            pol_dev = float(np.max(np.where(live, np.abs(act_t - ro.actions)/np.maximum(1., np.abs(ro.actions)), 0.))) if len(ro) else 0.
            same_len = len(ro) == len(ref)
            head = max_dev(ro[:3], ref[:3]) if same_len and len(ro) > 3 else (dev if same_len else np.inf)
            try:
                # (moved by 1e-13 and scaled to the unit roundoff: one ulp of a small initial angle is lost in the rounding
                # of the first observation it enters, cos(angle) ~ 1)
                ref2 = stepwise(build(rng, stage), init + 1e-13)
                sens = np.inf if len(ref2) != len(ref) else max_dev(ref2, ref)*(2.**-52/1e-13)
            except Exception:
                sens = np.inf
            # the environment alone: the fused rollout's recorded actions through the wrapper objects, step by step
            env_c = build(rng, stage)
            obs_c = [env_c.reset(init_state=init.copy(), domain_param=dict(dp_dict))]
            rew_c = []
            for a_k in np.asarray(ro.actions):
                o_k, r_k, _, _ = env_c.step(np.array(a_k, dtype=np.float64))
                obs_c.append(o_k)
                rew_c.append(r_k)
            # (rewards only of the steps whose recorded action is the policy's output: where QBallBalancerSim wrote its dead zone
            # into the action, the step reward was computed from the value before it)
            whole = np.all(np.asarray(ro.actions) != 0., axis=1)
            env_dev = max(cases.rel_err(np.asarray(ro.observations), np.stack(obs_c)),
                          cases.rel_err(np.asarray(ro.rewards)[whole], np.asarray(rew_c)[whole]))
            explained = (pol_dev <= 1e-12 and head <= 1e-12 and env_dev <= 1e-10) or dev <= 100*max(sens, 1e-12)
            bad += 0 if explained else 1
            print(f'[{seed}] {"(explained)" if explained else "MISMATCH"} fused={is_fused} lengths {len(ro)} / {len(ref)} dev {dev:.2e}; policy alone '
                  f'{pol_dev:.1e}, environment alone {env_dev:.1e}, first steps {head:.1e}, self-sensitivity {sens:.2e} | {what}', flush=True)
        # The same chain behind DomainRandWrapperLive in a ParallelSampler: the fused batch against the batched reset / step loop
        # that a policy the kernels cannot fuse takes (the same function, hidden in a plain torch module): the same domain
        # parameters and initial states are drawn on both paths, so lengths and returns must agree.
        if stage['stop'] and trial % 3 == 0:
            try:
                from simurlacra_amd.domain_randomization import get_default_randomizer
                from simurlacra_amd.environment_wrappers import DomainRandWrapperLive
                from simurlacra_amd.sampling import ParallelSampler

                class Opaque(to.nn.Module):
                    def forward(self, obs):
                        return pol(obs.to(to.float64))
                n_ro = int(rng.integers(3, 40))
                envs = [DomainRandWrapperLive(build(rng, stage), get_default_randomizer(build(rng, stage))) for _ in range(2)]
                a = ParallelSampler(envs[0], pol, 1, min_rollouts=n_ro, seed=seed).sample()
                b = ParallelSampler(envs[1], Opaque(), 1, min_rollouts=n_ro, seed=seed).sample()
                la, lb = np.asarray(a.lengths), np.asarray(b.lengths)
                ra, rb = np.asarray(a.undiscounted_returns()), np.asarray(b.undiscounted_returns())
                same = la == lb
                dev_s = float(np.max(np.abs(ra[same] - rb[same])/np.maximum(1., np.abs(rb[same])))) if same.any() else 0.
                if not same.all() or dev_s > 1e-6:
                    # (closed loops amplify: what counts is a deviation in the FIRST steps of a rollout)
                    k = int(np.flatnonzero(~same)[0]) if not same.all() else int(np.argmax(np.abs(ra - rb)))
                    head = max_dev(a[k][:2], b[k][:2]) if min(la[k], lb[k]) > 2 else 0.
                    ok = head <= 1e-12
                    bad += 0 if ok else 1
                    print(f'[{seed}] sampler {"(explained)" if ok else "MISMATCH"}: {int((~same).sum())} of {n_ro} lengths differ, returns dev {dev_s:.1e}, '
                          f'first steps of rollout {k}: {head:.1e} | {what}', flush=True)
                samplers += 1
            except Exception as ex:
                print(f'[{seed}] sampler not run ({type(ex).__name__}: {str(ex)[:140]}) | {what}', flush=True)
        if (trial + 1) % 50 == 0:
            print(f'... {trial + 1} trials', flush=True)
    return bad, fused, skipped, samplers


if __name__ == '__main__':
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad, fused, skipped, samplers = main(range(seed0, seed0 + trials), log=lambda m: print(m, flush=True))
    print(f'{trials} trials from seed {seed0}: {bad} unexplained mismatches, {fused} rollouts took the fused kernel, {skipped} chains not built, '
          f'{samplers} sampler pairs')
