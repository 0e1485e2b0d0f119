""" FNN population mode against the C oracle over random net shapes, rollouts per policy, population sizes and env
kinds (every layout of tiles per policy, waves per workgroup and tanh table the host can pick).
Usage: python tools/stress_population.py """
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import cases
from simurlacra_amd import _abi as abi
from simurlacra_amd.engine import get_engine
from oracle import oracle_lib as orc
eng = get_engine(0)
rng = np.random.default_rng(0)
kinds = [abi.ENV_QQ, abi.ENV_QCP_SU, abi.ENV_QCP_ST, abi.ENV_QBB, abi.ENV_BOB, abi.ENV_OMO, abi.ENV_PEND, abi.ENV_QQ_ST]
worst = 0.
for it in range(60):
    kind = kinds[it % len(kinds)]
    nh = int(rng.integers(1, 4))
    hidden = tuple(int(rng.choice([5, 16, 20, 32, 33, 64])) for _ in range(nh))
    R = int(rng.choice([1, 3, 8, 16, 17, 32, 40, 64, 65, 100]))
    pop = int(rng.choice([7, 100, 1500])) if R <= 17 else int(rng.choice([7, 100, 400]))
    T = 25
    case = cases.make_case(kind, pop*R, seed=100 + it, policy='fnn', hidden=hidden, scale=0.3)
    params = 0.3*rng.standard_normal((pop, case['ps'].num_param))
    ref = orc.rollout(kind, case['dt'], T, case['task_c'], case['ps'], params, case['init'], case['dp'], policy_group=R)
    out = eng.rollout(kind, case['dt'], T, case['task_c'], case['ps'], params, case['init'], case['dp'], policy_group=R)
    ln_bad = int((out['length'].cpu().numpy() != ref['length']).sum())
    st_bad = int((out['status'].cpu().numpy() != ref['status']).sum())
    err = cases.rel_err(out['ret'].cpu().numpy(), ref['ret'])
    worst = max(worst, err)
    flag = '' if (err <= 1e-9 and not ln_bad and not st_bad) else '  <<<<'
    print(f'{cases.KIND_NAMES[kind]:7s} hidden={hidden} pop={pop} R={R}: err {err:.2e} len-mismatch {ln_bad} status-mismatch {st_bad}{flag}', flush=True)
print('worst', worst)
