cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/pmc_insts.sh r02 > gpurun_out/r02_bench_instruction_mix.txt 2>&1 || { tail -20 gpurun_out/r02_bench_instruction_mix.txt; exit 1; }
cat gpurun_out/r02_bench_instruction_mix.txt
