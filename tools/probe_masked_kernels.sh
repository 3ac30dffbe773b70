#!/bin/bash
# tools/probe_masked_kernels.sh "t:0 t:32 u:32 b:32 ..."  -> per-mode kernel averages (rocprofv3 --kernel-trace --stats)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/masked
rm -rf $OUT; mkdir -p $OUT
for m in $1; do
  d=$OUT/$(echo $m | tr ':' '_')
  (cd /tmp && export TMPDIR=/tmp && PROBE_MODE=$m timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/tools/probe_masked_kernels.py > $d.log 2>&1) || echo "(mode $m: nonzero exit)"
  echo "== $m"
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("wino7_input", "wino7_output", "conv_igemm", "pw_tdiff16", "sobel_tdiff")):
        print("   %-70s calls %4s  avg %9.1f us  min %9.1f  max %9.1f" % (n[:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
