#!/bin/bash
# Profiles `bench.py` on the GPU box: kernel durations (rocprofv3 --kernel-trace --stats) and the HBM traffic of the
# rollout kernel (FETCH_SIZE and WRITE_SIZE in separate counter passes).  Usage: tools/profile_bench.sh <tag>
# Results land in gpurun_out/prof_<tag>/; summary.json there is what gets copied into profiles/.
set -e
tag=$1
out=gpurun_out/prof_$tag
mkdir -p $out
args="--steps 3 --warmup 1 --no-cpu-baseline"
timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d $out -o trace --output-format csv -- python3 bench.py $args > $out/trace.log 2>&1
timeout -s KILL 170 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out -o fetch --output-format csv -- python3 bench.py $args > $out/fetch.log 2>&1
timeout -s KILL 170 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out -o write --output-format csv -- python3 bench.py $args > $out/write.log 2>&1
python3 - $out <<'PY'
import csv, sys, json, collections
out = sys.argv[1]
def per_launch(name):
    vals = [float(r['Counter_Value']) for r in csv.DictReader(open(f'{out}/{name}_counter_collection.csv'))
            if 'k_rollout' in r['Kernel_Name']]
    by_dispatch = collections.defaultdict(float)
    for r in csv.DictReader(open(f'{out}/{name}_counter_collection.csv')):
        if 'k_rollout' in r['Kernel_Name']:
            by_dispatch[r['Dispatch_Id']] += float(r['Counter_Value'])
    v = list(by_dispatch.values())
    return sum(v)/len(v), len(v)
fetch_kb, n1 = per_launch('fetch')
write_kb, n2 = per_launch('write')
stats = [r for r in csv.DictReader(open(f'{out}/trace_kernel_stats.csv')) if 'k_rollout' in r['Name']]
bench = None
for line in open(f'{out}/trace.log'):
    if line.startswith('{"metric"'):
        bench = json.loads(line)
summary = dict(fetch_size_kb_per_launch=fetch_kb, write_size_kb_per_launch=write_kb, launches=[n1, n2],
               hbm_bytes_per_launch=(2*fetch_kb + write_kb)*1024.,
               note='FETCH_SIZE doubled (gfx950 tallies 128-B read requests at 64 B); 8-B-per-lane accesses are uncalibrated',
               kernel_stats=stats, bench_under_rocprof=bench)
json.dump(summary, open(f'{out}/summary.json', 'w'), indent=1)
print(json.dumps(summary, indent=1))
PY
