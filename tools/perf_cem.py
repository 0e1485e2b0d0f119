"""One CEM generation of LinearPolicies on QBallBalancerSim (the measurement plan's C4: 4 096 parameter sets x 8 shared
rollouts, T = 500, default randomizer) with and without the fused wrappers, to see what each wrapper costs a launch
that is latency bound (one wave per SIMD at most).  Usage: python tools/perf_cem.py [pop] [R] [linear|16|32|64]
(a number: FNNPolicy with two tanh layers of that width instead of the LinearPolicy)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simurlacra_amd import _abi as abi, envdefs  # noqa: E402
from simurlacra_amd.domain_randomization.default_randomizers import get_default_randomizer  # noqa: E402
from simurlacra_amd.domain_randomization.domain_parameter import UniformDomainParam  # noqa: E402
from simurlacra_amd.engine import get_engine  # noqa: E402
from simurlacra_amd.environments.pysim import QBallBalancerSim, QQubeSim  # noqa: E402
from simurlacra_amd.specs import PolicySpec, WrapSpec  # noqa: E402

pop = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
R = int(sys.argv[2]) if len(sys.argv) > 2 else 8
pol = sys.argv[3] if len(sys.argv) > 3 else 'linear'
eng = get_engine(0)
dev = eng.device


def generation(env, dt, T, tag, wrap_ops, delay_row, general_task=False):
    kind = env.kind
    S, O, A, P, _ = abi.ENV_DIMS[kind]
    rand = get_default_randomizer(env)
    rand.add_domain_params(UniformDomainParam(name='act_delay', mean=5, halfspan=5, clip_lo=0, roundint=True))
    nominal = list(env.domain_param_column()) + [0.]
    dp_r = eng.randomize(R, rand.device_table(envdefs.DP_NAMES[kind], ['act_delay']), nominal, seed=3)
    init_r = eng.sample_init(kind, R, dp_r, seed=3)
    g = torch.Generator().manual_seed(0)
    if pol == 'linear':
        ps = PolicySpec('linear', O, A, feats=['identity'])
        theta0 = 0.1*torch.randn(ps.num_param, dtype=torch.float64, generator=g)
        theta = (theta0 + 0.5*torch.randn((pop, ps.num_param), dtype=torch.float64, generator=g)).to(dev)
    else:
        ps = PolicySpec('fnn', O, A, hidden=[int(pol)]*2, hidden_nonlin=['tanh', 'tanh'])
        theta = (0.2*torch.randn((pop, ps.num_param), dtype=torch.float64, generator=g)).to(dev)
    r_idx = torch.arange(R, dtype=torch.int32, device=dev).repeat(pop)
    init = init_r[:, r_idx.long()].contiguous()
    wrap = None
    if wrap_ops:
        wrap = WrapSpec(ops=wrap_ops, act_delay_row=P if delay_row else -1, act_delay=0 if delay_row else 5)
    kw = dict(final_state=False, dp_index=r_idx, policy_group=R)
    if wrap is not None:
        kw['wrap'] = wrap
    ts = env.task_spec()
    if general_task:  # a coupling of no numerical weight: the task no longer has the simulation's default shape
        ts.Q = ts.Q.copy()
        ts.Q[0, 1] = ts.Q[1, 0] = 1e-300
    for _ in range(2):
        out = eng.rollout(kind, dt, T, ts.to_c(S, A), ps, theta, init, dp_r, **kw)
    ms = eng.last_rollout_ms()
    steps = int(out['length'].sum().item())
    print(f'{tag:64s} N={pop*R:7d} mean_len={steps/(pop*R):6.1f} max_len={int(out["length"].max()):4d} kernel {ms:7.3f} ms '
          f'{steps/ms/1e6:7.3f} G env-steps/s', flush=True)


for env, dt, T, name in ((QBallBalancerSim(dt=0.01, max_steps=500), 0.01, 500, 'QBB'),
                         (QQubeSim(dt=1/500., max_steps=600), 1/500., 600, 'QQ')):
    generation(env, dt, T, f'{name} pop={pop} x R={R} {pol}, no wrapper', [], False)
    generation(env, dt, T, f'{name} same, a task without the default shape (general kernel)', [], False, True)
    generation(env, dt, T, f'{name} same, ActNormWrapper', ['act_norm'], False)
    generation(env, dt, T, f'{name} same, ActDelayWrapper(5)', ['act_delay'], False)
    generation(env, dt, T, f'{name} same, ActDelayWrapper, act_delay 0..10 per rollout', ['act_delay'], True)
    generation(env, dt, T, f'{name} same, ActNorm(ActDelay), act_delay 0..10 per rollout', ['act_norm', 'act_delay'], True)
