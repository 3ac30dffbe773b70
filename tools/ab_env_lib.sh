#!/bin/bash
# same-box sweep of one environment knob of a tuning build: tools/ab_env_lib.sh <lib> VAR "v1 v2 ..." [rounds] [time_forward args...]
LIB=$1; VAR=$2; VALS=$3; REP=${4:-3}; shift 4
for r in $(seq 1 $REP); do
  for v in $VALS; do
    echo "$VAR=$v"
    env OFFK_LIB=$LIB $VAR=$v timeout -k 10 200 python tools/time_forward.py "$@" 2>/dev/null
  done
done
