cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
timeout -k 10 400 python -u tools/stress_population.py > gpurun_out/r02_stress_population.txt 2>&1 || { tail gpurun_out/r02_stress_population.txt; exit 1; }
tail -3 gpurun_out/r02_stress_population.txt; grep -c "<<<<" gpurun_out/r02_stress_population.txt
