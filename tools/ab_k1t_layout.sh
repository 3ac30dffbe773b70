#!/bin/bash
# same-box A/B of the K1T block layout knob of a tuning build (OFFK_PW_STREAM: the 7x7 sites' pixels as a stream of quads):
#   tools/ab_k1t_layout.sh <liboffk_knobs.so> [rounds]       (profiles/r04/ab_k1t_quad_stream.txt also swept the site order, since removed)
LIB=$1; REP=${2:-3}
for r in $(seq 1 $REP); do
  for v in 0 1; do
    echo "stream=$v"
    OFFK_LIB=$LIB OFFK_PW_STREAM=$v timeout -k 10 200 python tools/time_forward.py 64 7 100 "pw_tdiff" 2>/dev/null
  done
done
