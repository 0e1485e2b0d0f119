#!/bin/bash
# The round's closing pass on the GPU box: test suite, stress sweeps, the wrapper / population / configuration tables and
# the bench profile, everything into gpurun_out/.  Usage: gpurun -- bash tools/final_profiles.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
set -e
timeout -k 10 600 python -u -m pytest tests -m gpu -x -v --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -1 gpurun_out/gpu_tests.log
timeout -k 10 800 python -u tools/stress_parity.py 25 > gpurun_out/r03_stress_parity_now.txt 2>/dev/null
tail -1 gpurun_out/r03_stress_parity_now.txt
timeout -k 10 400 python -u tools/stress_population.py 2>/dev/null > gpurun_out/r03_stress_population_now.txt
tail -1 gpurun_out/r03_stress_population_now.txt
for p in linear 16 32 64; do timeout -k 10 200 python -u tools/perf_cem.py 4096 8 $p 2>/dev/null; done > gpurun_out/perf_cem_after.txt
echo cem done
{ timeout -k 10 300 python -u tools/perf_wrapped.py 1048576 fnn 2>/dev/null; timeout -k 10 300 python -u tools/perf_wrapped.py 1048576 linear 2>/dev/null; } > gpurun_out/perf_wrapped_after.txt
echo wrapped done
timeout -k 10 400 python -u tools/bench_configs.py > gpurun_out/r03_bench_configs_now.txt 2>gpurun_out/bench_configs.err
echo configs done
timeout -k 10 250 python -u tools/perf_population.py > gpurun_out/r03_population_now.txt 2>/dev/null
echo population done
bash tools/profile_bench.sh r03final > gpurun_out/profile_bench_r03final.txt 2>&1
bash tools/pmc_insts.sh r03 > gpurun_out/r03_bench_instruction_mix.txt 2>&1
echo insts done
echo profile done
timeout -k 10 300 python -u bench.py > gpurun_out/bench_now.json 2> gpurun_out/bench_now.err
cut -c1-330 gpurun_out/bench_now.json
