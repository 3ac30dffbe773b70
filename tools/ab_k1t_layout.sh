#!/bin/bash
# same-box A/B of the K1T block layout knobs (tuning build): tools/ab_k1t_layout.sh <liboffk_knobs.so> [rounds]
LIB=$1; REP=${2:-3}
for r in $(seq 1 $REP); do
  for cfg in "0 0" "1 0" "0 1" "1 1"; do
    set -- $cfg
    echo "order=$1 stream=$2"
    OFFK_LIB=$LIB OFFK_PW_ORDER=$1 OFFK_PW_STREAM=$2 timeout -k 10 200 python tools/time_forward.py 64 7 100 "pw_tdiff" 2>/dev/null
  done
done
