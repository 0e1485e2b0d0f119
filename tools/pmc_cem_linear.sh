#!/bin/bash
# Dynamic instruction counts of the LinearPolicy population launches of tools/perf_cem.py, per kernel and launch.
set -e
out=gpurun_out/pmc_cem_linear
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 170 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_SALU SQ_INSTS_LDS -d $out -o a --output-format csv -- python3 tools/perf_cem.py > $out/a.log 2>&1
timeout -s KILL 170 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d $out -o b --output-format csv -- python3 tools/perf_cem.py > $out/b.log 2>&1
timeout -s KILL 170 rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_WAIT_ANY -d $out -o c --output-format csv -- python3 tools/perf_cem.py > $out/c.log 2>&1
python3 - $out <<'PY'
import csv, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
for f in ('a', 'b', 'c'):
    for r in csv.DictReader(open(f'{out}/{f}_counter_collection.csv')):
        name = r['Kernel_Name']
        if 'k_rollout' in name:
            agg[name][r['Counter_Name']] += float(r['Counter_Value'])
            launches[(name, f)].add(r['Dispatch_Id'])
for name in sorted(agg):
    nl = max(1, len(launches[(name, 'a')]))
    print(name, 'launches', nl)
    for k, v in sorted(agg[name].items()):
        print(f'   {k:28s} {v/nl:16.0f} per launch')
PY
