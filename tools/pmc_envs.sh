#!/bin/bash
# Dynamic instruction counts of the LinearPolicy rollout kernels (tools/perf_envs.py with zero weights, so that the
# rollouts run their full horizon): three counter passes, per-kernel sums divided by the number of launches.
# Usage: tools/pmc_envs.sh <tag> [n_envs] [linear|fnn] [features of the LinearPolicy, e.g. identity]
set -e
out=gpurun_out/pmc_envs_$1
n=${2:-1048576}
pol=${3:-linear}
feats=${4:-}
mkdir -p $out
timeout -s KILL 170 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_SALU SQ_INSTS_LDS -d $out -o a --output-format csv -- python3 tools/perf_envs.py $n $pol 0.0 $feats > $out/a.log 2>&1
timeout -s KILL 170 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d $out -o b --output-format csv -- python3 tools/perf_envs.py $n $pol 0.0 $feats > $out/b.log 2>&1
timeout -s KILL 170 rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC -d $out -o c --output-format csv -- python3 tools/perf_envs.py $n $pol 0.0 $feats > $out/c.log 2>&1
python3 - $out $pol <<'PY'
import csv, sys, collections, re
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
for f in ('a', 'b', 'c'):
    for r in csv.DictReader(open(f'{out}/{f}_counter_collection.csv')):
        name = r['Kernel_Name']
        if 'k_rollout' in name:
            agg[name][r['Counter_Name']] += float(r['Counter_Value'])
            launches[(name, f)].add(r['Dispatch_Id'])
steps = {}
for line in open(f'{out}/a.log'):
    m = re.match(r'(\w+)\s+%s\s+n=(\d+) mean_len=\s*([\d.]+)' % sys.argv[2], line)
    if m:
        steps[m.group(1)] = float(m.group(2))*float(m.group(3))
print('env steps per launch:', steps)
for name in sorted(agg):
    kind = re.search(r'<(\d+)', name)
    key = ['QQ', 'QCP_SU', 'QCP_ST', 'QBB', 'BOB', 'OMO', 'PEND', 'QQ_ST'][int(kind.group(1))] if kind else None
    ws = steps.get(key, 0.)/64.
    print(name, f'wave-steps per launch {ws:.0f}')
    for k, v in sorted(agg[name].items()):
        nl = max(1, len(launches[(name, 'a')]))   # every pass runs the same launches
        print(f'   {k:28s} {v/nl:16.0f} per launch {v/nl/ws if ws else 0.:10.1f} per wave-step')
PY
