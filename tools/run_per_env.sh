cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1; tail -3 gpurun_out/gpu_tests.log
{ echo "# python tools/perf_envs.py 1048576 linear 0.5   (FeatureStack const+identity+squared, the setting of r01_per_env_kernels.txt)"
  timeout -k 10 200 python -u tools/perf_envs.py 1048576 linear 0.5 2>/dev/null
  echo "# python tools/perf_envs.py 1048576 linear 0.0 identity   (identity features, zero weights: full-length rollouts; kernel k_rollout_linear<KIND, false, false, SIMPLE, FIXT>)"
  timeout -k 10 200 python -u tools/perf_envs.py 1048576 linear 0.0 identity 2>/dev/null
  echo "# python tools/perf_envs.py 1048576 linear 0.5 identity"
  timeout -k 10 200 python -u tools/perf_envs.py 1048576 linear 0.5 identity 2>/dev/null
  echo "# python tools/perf_envs.py 1048576 fnn 0.5"
  timeout -k 10 200 python -u tools/perf_envs.py 1048576 fnn 0.5 2>/dev/null
} > gpurun_out/r02_per_env_kernels.txt
cat gpurun_out/r02_per_env_kernels.txt
