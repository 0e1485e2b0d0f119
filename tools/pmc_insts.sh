#!/bin/bash
# Dynamic instruction mix of the bench kernel: two counter passes (at most 8 SQ counters each), one launch each.
set -e
out=gpurun_out/pmc_$1
mkdir -p $out
timeout -s KILL 170 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT -d $out -o a --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/a.log 2>&1
timeout -s KILL 170 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS -d $out -o b --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/b.log 2>&1
python3 - $out <<'PY'
import csv, sys, collections, json
out = sys.argv[1]
agg = collections.defaultdict(float)
for f in ('a', 'b'):
    for r in csv.DictReader(open(f'{out}/{f}_counter_collection.csv')):
        if 'k_rollout' in r['Kernel_Name']:
            agg[r['Counter_Name']] += float(r['Counter_Value'])
steps = None
for line in open(f'{out}/a.log'):
    if line.startswith('{"metric"'):
        steps = json.loads(line)['roofline']['env_steps_per_launch']
print('env_steps_per_launch', steps)
ws = steps/64.
for k, v in sorted(agg.items()):
    print(f'{k:32s} {v:16.0f}  per wave-step {v/ws:10.1f}')
PY
