cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || { tail -40 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
echo "== copies"; timeout -k 10 300 python -u tools/perf_batch.py 2>/dev/null | head -6 || exit 1
cp tools/var/lib_head.so simurlacra_amd/csrc/libpysimhip.so
echo "== head"; timeout -k 10 300 python -u tools/perf_batch.py 2>/dev/null | head -6 || exit 1
