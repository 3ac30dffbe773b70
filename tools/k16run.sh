#!/bin/bash
# A/B of liboffk builds (tools/build_variant.py) on one box: tools/k16run.sh reps lib...
REP=$1; shift
for r in $(seq 1 $REP); do
for l in "$@"; do OFFK_LIB=$l timeout -k 10 150 python bench.py --no-secondary --steps 100 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$l', round(d['ms_per_step'], 4), round(d['stage_ms']['pw_reduce'], 4))"; done; done
