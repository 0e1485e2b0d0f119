""" Writes the episode lengths of one launch of the bench workload (env order) to gpurun_out/lengths.npy, the input of
tools/tail_model.py.  Usage: python tools/dump_lengths.py [sets] """
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench                                                                                           # noqa: E402
from simurlacra_amd import _abi as abi, envdefs                                                        # noqa: E402
from simurlacra_amd.engine import get_engine                                                           # noqa: E402
from simurlacra_amd.specs import PolicySpec                                                            # noqa: E402

n_sets = int(sys.argv[1]) if len(sys.argv) > 1 else 8
R = 262144
eng = get_engine(0)
S, O, A, P, _ = abi.ENV_DIMS[abi.ENV_QQ]
task_c = envdefs.default_task_spec(abi.ENV_QQ).to_c(S, A)
ps = PolicySpec('fnn', O, A, hidden=[64, 64], hidden_nonlin=['tanh', 'tanh'])
params = bench.reference_fnn_params([O, 64, 64, A], seed=0).to(eng.device)
N = n_sets*R
dp = torch.empty((P, N), dtype=torch.float64, device=eng.device)
for j in range(n_sets):
    distrs, nominal = bench.qq_set_distrs(j)
    dp[:, j*R:(j + 1)*R] = eng.randomize(R, distrs, nominal, seed=0, env_offset=j*R)
init = eng.sample_init(abi.ENV_QQ, N, dp, seed=0, env_offset=0)
out = eng.rollout(abi.ENV_QQ, bench.DT, bench.T, task_c, ps, params, init, dp, final_state=False)
torch.cuda.synchronize()
ln = out['length'].cpu().numpy().astype(np.int16)
os.makedirs('gpurun_out', exist_ok=True)
np.save('gpurun_out/lengths.npy', ln)
print('kernel ms', eng.last_rollout_ms(), 'mean length', ln.mean(), 'share at T', (ln == bench.T).mean())
print('deciles', np.percentile(ln, np.arange(0, 101, 10)))
