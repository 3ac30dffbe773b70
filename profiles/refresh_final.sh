#!/bin/bash
# End-of-round evidence in one gpurun call: profiles/refresh_final.sh <tag>  -> gpurun_out/<tag>/
#   bench.json (the driver's command) + bench_detail.json, rocprofv3 kernel stats of the split-fp32 (headline) and fp32-pipe forwards, PMC passes, batch table.
set -u
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 --detail $OUT/bench_detail.json > $OUT/bench.json 2> $OUT/bench.err || exit 1
echo "bench done"
# the headline mode (split-fp32: the default of bench.py since round 6)
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_f32split -- python3 $GRAFT_REPO_ROOT/bench.py --no-secondary --precision f32split --steps 100 --warmup 10 --detail $OUT/detail_stats_f32split.json > $OUT/stats_f32split.log 2>&1) || exit 1
echo "split stats done"
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-secondary --precision fp32 --steps 100 --warmup 10 --detail $OUT/detail_stats_fp32.json > $OUT/stats.log 2>&1) || exit 1
echo "stats done"
bash profiles/collect_pmc.sh $TAG/pmc_f32split --precision f32split --detail /tmp/d.json || exit 1
bash profiles/collect_pmc.sh $TAG/pmc --precision fp32 --detail /tmp/d.json || exit 1
timeout -k 10 300 python tools/bench_small.py 1,7 8,7 16,7 42,7 55,7 128,7 10,25 --fp32 > $OUT/batch_table.txt 2>&1 || exit 1
timeout -k 10 300 python tools/bench_small.py 1,7 8,7 16,7 42,7 55,7 128,7 10,25 --f32split > $OUT/batch_table_f32split.txt 2>&1 || exit 1
echo "table done"
