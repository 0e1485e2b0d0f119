cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method thread -k "edges_of_their_layouts" > gpurun_out/edge_tests.txt 2>&1 || { tail -40 gpurun_out/edge_tests.txt; exit 1; }
tail -2 gpurun_out/edge_tests.txt
timeout -k 10 300 python -u tools/perf_population.py 2>/dev/null > gpurun_out/r02_population.txt || exit 1
timeout -k 10 300 python -u tools/perf_batch.py 2>/dev/null > gpurun_out/batch_new.txt || exit 1
timeout -k 10 300 python -u tools/perf_batch.py 16,16 2>/dev/null > gpurun_out/batch_new16.txt || exit 1
cat gpurun_out/r02_population.txt gpurun_out/batch_new.txt
