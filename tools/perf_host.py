#!/usr/bin/env python3
""" Wall time of the samplers as a caller sees them (Python layer + launches + result objects) next to the time of the
rollout kernel alone: ParallelSampler.sample() on randomized QQube with an FNN policy, with and without trajectories,
and ParameterExplorationSampler.sample() on the CEM workload (QBallBalancer, 4096 LinearPolicy sets x 8 rollouts).
Usage: python tools/perf_host.py """
import os
import sys
import time

import torch as to

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simurlacra_amd.domain_randomization.default_randomizers import get_default_randomizer  # noqa: E402
from simurlacra_amd.engine import get_engine  # noqa: E402
from simurlacra_amd.environment_wrappers.domain_randomization import DomainRandWrapperLive  # noqa: E402
from simurlacra_amd.environments.pysim.quanser_ball_balancer import QBallBalancerSim  # noqa: E402
from simurlacra_amd.environments.pysim.quanser_qube import QQubeSim  # noqa: E402
from simurlacra_amd.policies.features import FeatureStack, identity_feat  # noqa: E402
from simurlacra_amd.policies.fnn import FNNPolicy  # noqa: E402
from simurlacra_amd.policies.linear import LinearPolicy  # noqa: E402
from simurlacra_amd.sampling.parallel_sampler import ParallelSampler  # noqa: E402
from simurlacra_amd.sampling.parameter_exploration_sampler import ParameterExplorationSampler  # noqa: E402

eng = get_engine(0)


def timed(fn, reps=3):
    fn()
    to.cuda.synchronize()
    best, kernel = 1e9, 0.
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        to.cuda.synchronize()
        el = time.perf_counter() - t0
        if el < best:
            best, kernel = el, eng.last_rollout_ms()
    return best*1e3, kernel, out


sim = QQubeSim(dt=1/500., max_steps=600)
env = DomainRandWrapperLive(sim, get_default_randomizer(sim))
policy = FNNPolicy(env.spec, hidden_sizes=[64, 64], hidden_nonlin=to.tanh)
for n, rec in ((1000, True), (100000, False), (100000, True), (1000000, False)):
    sampler = ParallelSampler(env, policy, num_envs=1, min_rollouts=n, seed=0, record_trajectories=rec)
    ms, k, ros = timed(sampler.sample)
    t0 = time.perf_counter()
    rets = ros.undiscounted_returns()
    t_ret = (time.perf_counter() - t0)*1e3
    t0 = time.perf_counter()
    first = ros[0]
    t_first = (time.perf_counter() - t0)*1e3
    print(f'ParallelSampler QQube FNN 64x64 n={n:8d} trajectories={rec!s:5s}: sample() {ms:8.2f} ms, rollout kernel {k:8.2f} ms, '
          f'undiscounted_returns() {t_ret:6.2f} ms, ros[0] {t_first:6.2f} ms, mean return {rets.mean():.3f}, len(ros[0]) {len(first)}', flush=True)

sim = QBallBalancerSim(dt=0.01, max_steps=500)
env = DomainRandWrapperLive(sim, get_default_randomizer(sim))
policy = LinearPolicy(env.spec, FeatureStack([identity_feat]))
pes = ParameterExplorationSampler(env, policy, num_envs=1, num_rollouts_per_param=8, seed=1, record_trajectories=False)
theta = 0.05*to.randn(4096, policy.num_param, dtype=to.float64, generator=to.Generator().manual_seed(0))
ms, k, res = timed(lambda: pes.sample(theta))
t0 = time.perf_counter()
idx, val, par = res.elites(256)
t_el = (time.perf_counter() - t0)*1e3
print(f'ParameterExplorationSampler QBallBalancer 4096 x 8 LinearPolicy: sample() {ms:8.2f} ms, rollout kernel {k:8.2f} ms, '
      f'elites(256) {t_el:6.2f} ms, best mean return {val[0]:.3f}', flush=True)
