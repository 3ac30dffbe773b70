#!/bin/bash
# Vector-memory path counters of one short bench run (what stalls the units kernel's loads): profiles/collect_pmc_vmem.sh <outdir-under-gpurun_out> [bench args]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do      # (a third pass with the TA_* counters aborted inside rocprofv3 on this pool and sat out the silence timer: not collected)
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-secondary "$@" > $OUT/$tag.log 2>&1
  echo "$tag rc=$?"
done
