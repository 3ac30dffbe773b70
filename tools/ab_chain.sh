#!/bin/bash
# same-box A/B of the chain launches in split-fp32 arithmetic: tools/ab_chain.sh "libA libB ..." [rounds]
LIBS=$1; REP=${2:-2}
export OFFK_PRECISION=f32split
for r in $(seq 1 $REP); do
  for l in $LIBS; do
    OFFK_LIB=$PWD/$l timeout -k 10 200 python tools/time_forward.py 64 7 100 "chain_" 2>/dev/null | grep -v "sum of" | sed 's/= motion.*merged_28a\|= motion_conv1.*trans_28[bc] *//' | tr '\n' ' ' | sed 's/ \+/ /g'; echo
  done
done
