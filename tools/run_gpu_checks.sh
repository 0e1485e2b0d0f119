#!/bin/bash
# One GPU-box call: the GPU test suite, bench.py, the instruction counters of the LinearPolicy kernels and the per-simulation
# kernel table (gpurun_out/r02_per_env_kernels.txt).  Usage: gpurun -- bash tools/run_gpu_checks.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
timeout -k 10 300 python -u bench.py > gpurun_out/bench_now.json 2> gpurun_out/bench_now.err || { tail gpurun_out/bench_now.err; exit 1; }
cat gpurun_out/bench_now.json
bash tools/pmc_envs.sh r02_identity 1048576 linear identity > gpurun_out/pmc_envs_r02_identity.txt 2>&1 || { tail gpurun_out/pmc_envs_r02_identity.txt; exit 1; }
{ echo "# python tools/perf_envs.py 1048576 linear 0.5   (FeatureStack const+identity+squared, the setting of r01_per_env_kernels.txt)"
  timeout -k 10 200 python -u tools/perf_envs.py 1048576 linear 0.5 2>/dev/null
  echo "# python tools/perf_envs.py 1048576 linear 0.0 identity   (identity features, zero weights: full-length rollouts; kernel k_rollout_linear<KIND, false, false, SIMPLE, FIXT>)"
  timeout -k 10 200 python -u tools/perf_envs.py 1048576 linear 0.0 identity 2>/dev/null
  echo "# python tools/perf_envs.py 1048576 linear 0.5 identity"
  timeout -k 10 200 python -u tools/perf_envs.py 1048576 linear 0.5 identity 2>/dev/null
  echo "# python tools/perf_envs.py 1048576 fnn 0.5"
  timeout -k 10 200 python -u tools/perf_envs.py 1048576 fnn 0.5 2>/dev/null
} > gpurun_out/r02_per_env_kernels.txt
