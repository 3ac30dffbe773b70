#!/bin/bash
# A/B of an environment switch read at offk_create: tools/ab_env.sh VAR "v1 v2 ..." [repeats] [bench args...]
VAR=$1; VALS=$2; REP=${3:-2}; shift 3
for r in $(seq 1 $REP); do
  for v in $VALS; do
    env $VAR=$v timeout -k 10 150 python bench.py --no-secondary --steps 100 --warmup 10 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$VAR=$v', round(d['ms_per_step'], 4), {k: round(x, 4) for k, x in d['stage_ms'].items()})"
  done
done
