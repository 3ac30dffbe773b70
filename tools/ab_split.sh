#!/bin/bash
# same-box A/B of the units kernel in split-fp32 arithmetic: tools/ab_split.sh "libA libB ..." [rounds] ; OFFK_SPLIT_PC passes through
LIBS=$1; REP=${2:-2}
export OFFK_PRECISION=f32split
for r in $(seq 1 $REP); do
  for l in $LIBS; do
    for pc in ${PCS:-1}; do
      OFFK_SPLIT_PC=$pc OFFK_LIB=$PWD/$l timeout -k 10 200 python tools/time_forward.py 64 7 100 "K1T" 2>/dev/null | grep -v "sum of" | tr '\n' ' ' | sed 's/ \+/ /g'; echo " pc=$pc"
    done
  done
done
