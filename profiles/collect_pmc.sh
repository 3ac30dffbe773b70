#!/bin/bash
# Collects per-kernel PMC counters for one short bench run (separate passes, as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE cannot share a pass).
# usage: profiles/collect_pmc.sh <outdir-under-gpurun_out> [extra bench args]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-secondary "$@" > $OUT/$tag.log 2>&1
  echo "$tag rc=$?"
done
