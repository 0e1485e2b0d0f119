cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_host_api.py -x -q --timeout 120 --timeout-method thread -k "popul or config or param_expl" > gpurun_out/pop_tests.txt 2>&1 || { tail -40 gpurun_out/pop_tests.txt; exit 1; }
tail -2 gpurun_out/pop_tests.txt
echo "== spread"; timeout -k 10 300 python -u tools/perf_population.py 2>/dev/null | grep -E "pop=(128|256|1024) " || exit 1
cp tools/var/lib_head.so simurlacra_amd/csrc/libpysimhip.so
echo "== head"; timeout -k 10 300 python -u tools/perf_population.py 2>/dev/null | grep -E "pop=(128|256|1024) " || exit 1
