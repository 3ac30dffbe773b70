#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/gemm_rounds
rm -rf $OUT; mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/probe_gemm_rounds.py > $OUT/log.txt 2>&1)
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, os
ns = [int(v) for v in os.environ.get("PROBE_NS", "256,320,384,448,512,576,640,704,768").split(",")]
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "conv_igemm" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print("launches", len(d))
for i, n in enumerate(ns):
    v = d[4 * i:4 * i + 4]
    gx = int(rows[4 * i]["Grid_Size_X"]) // 256 if "Grid_Size_X" in rows[0] else -1
    gy = int(rows[4 * i]["Grid_Size_Y"]) if "Grid_Size_Y" in rows[0] else -1
    blocks = gx * gy
    print("n = %4d  grid %d x %d = %5d blocks = %.2f rounds of 1280:  %s us   (min %.1f, per block-round %.1f)" % (n, gx, gy, blocks, blocks / 1280.0, " ".join("%.1f" % t for t in v), min(v), min(v) / (blocks / 1280.0)))
PY
