#!/bin/bash
# Copies what tools/final_profiles.sh left under gpurun_out/ into profiles/ under this round's names.
# Usage: bash tools/collect_profiles.sh r03
set -e
R=${1:-r03}
G=gpurun_out
P=profiles
cp $G/prof_${R}final/trace_kernel_stats.csv $P/${R}_bench_kernel_stats.csv
cp $G/prof_${R}final/summary.json $P/${R}_bench_summary.json
cp $G/${R}_bench_instruction_mix.txt $P/${R}_bench_instruction_mix.txt
cp $G/${R}_bench_configs_now.txt $P/${R}_bench_configs.txt
cp $G/${R}_population_now.txt $P/${R}_population.txt
cp $G/${R}_stress_parity_now.txt $P/${R}_stress_parity.txt
cp $G/${R}_stress_population_now.txt $P/${R}_stress_population.txt
cp $G/perf_cem_after.txt $P/${R}_cem.txt
cp $G/perf_wrapped_after.txt $P/${R}_wrapped.txt
cp $G/bench_now.json $P/${R}_bench.json
git -C . rev-parse --short HEAD > $P/${R}_commit.txt
