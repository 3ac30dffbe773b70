#!/bin/bash
# same-box A/B of the Winograd GEMMs of a split-fp32 handle: fp32 pipe (OFFK_SPLIT_GEMM=0) vs split-fp32 (1): tools/ab_split_gemm.sh [rounds] [lib]
REP=${1:-2}
export OFFK_PRECISION=f32split
[ -n "$2" ] && export OFFK_LIB=$PWD/$2
for r in $(seq 1 $REP); do
  for sg in 0 1; do
    OFFK_SPLIT_GEMM=$sg timeout -k 10 200 python tools/time_forward.py 64 7 100 "GEMMs" 2>/dev/null | grep -v "sum of" | tr '\n' ' ' | sed 's/ \+/ /g'; echo " split_gemm=$sg"
  done
done
