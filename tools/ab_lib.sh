#!/bin/bash
# same-box A/B of library builds: tools/ab_lib.sh "libA.so libB.so" [rounds] [time_forward args...]
LIBS=$1; REP=${2:-3}; shift 2
for r in $(seq 1 $REP); do
  for l in $LIBS; do
    OFFK_LIB=$l timeout -k 10 200 python tools/time_forward.py "$@" 2>/dev/null
  done
done
