#!/bin/bash
# Dynamic instruction counts of the FNN population launches of tools/perf_population.py <hidden>, per kernel and launch.
# Usage: tools/pmc_population.sh 64,64
set -e
out=gpurun_out/pmc_population
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 170 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_SALU SQ_INSTS_LDS -d $out -o a --output-format csv -- python3 tools/perf_population.py $1 > $out/a.log 2>&1
timeout -s KILL 170 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_WAVES -d $out -o b --output-format csv -- python3 tools/perf_population.py $1 > $out/b.log 2>&1
python3 - $out <<'PY'
import csv, sys, collections
out = sys.argv[1]
rows = collections.defaultdict(lambda: collections.defaultdict(float))
names = {}
for f in ('a', 'b'):
    order = []
    for r in csv.DictReader(open(f'{out}/{f}_counter_collection.csv')):
        if 'k_rollout' in r['Kernel_Name']:
            d = int(r['Dispatch_Id'])
            if d not in order:
                order.append(d)
            rows[order.index(d)][r['Counter_Name']] += float(r['Counter_Value'])
            names[order.index(d)] = r['Kernel_Name']
lines = [l for l in open(f'{out}/a.log') if 'kernel' in l and 'pop=' in l]
for i in sorted(rows):
    if i % 2 == 0:   # every configuration is launched twice
        continue
    print(lines[i//2].strip() if i//2 < len(lines) else '?')
    print('   ', names[i])
    c = rows[i]
    print('    ' + '  '.join(f'{k[3:]}={v:.3g}' for k, v in sorted(c.items())))
PY
