#!/bin/bash
# HBM bytes of the trajectory-recording rollout kernels (tools/perf_writeback.py): WRITE_SIZE and FETCH_SIZE in counter
# passes of their own, per launch of each k_rollout kernel, next to the algorithmic bytes the tool prints.
# Usage: tools/pmc_writeback.sh <tag> [n_envs] [kinds] [time_major|env_major]
set -e
out=gpurun_out/pmc_wb_$1
n=${2:-2097152}
kinds=${3:-omo,bob}
mode=${4:-time_major}
mkdir -p $out
timeout -s KILL 170 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out -o w --output-format csv -- python3 tools/perf_writeback.py $n $kinds $mode > $out/w.log 2>&1
timeout -s KILL 170 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out -o f --output-format csv -- python3 tools/perf_writeback.py $n $kinds $mode > $out/f.log 2>&1
python3 - $out <<'PY'
import csv, sys, collections
out = sys.argv[1]
for tag, name in (('w', 'WRITE_SIZE'), ('f', 'FETCH_SIZE')):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f'{out}/{tag}_counter_collection.csv')):
        if 'k_rollout' in r['Kernel_Name']:
            per[r['Kernel_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
    for k, d in per.items():
        v = list(d.values())
        print(f'{name} {k}: {len(v)} launches, {sum(v)/len(v)*1024/1e9:.3f} GB per launch (counter in KB x 1024; FETCH_SIZE not doubled)')
print(open(f'{out}/w.log').read())
PY
