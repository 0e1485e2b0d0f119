#!/bin/bash
# A/B runs of bench.py over variant builds of the library (tools/var/lib_<tag>.so), on the GPU box's scratch copy of
# the tree: each variant is copied over simurlacra_amd/csrc/libpysimhip.so in turn, twice, interleaved.
# Usage: tools/ab_bench.sh <tag> [<tag> ...]
set -e
for round in 1 2; do
    for tag in "$@"; do
        cp tools/var/lib_$tag.so simurlacra_amd/csrc/libpysimhip.so
        timeout -k 10 120 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$tag', round(d['ms_per_step'], 2), 'ms', '%.4g' % d['value'], 'kernel', round(d['roofline']['kernel_ms'], 2), 'frac', round(d['roofline']['frac'], 4))"
    done
done
