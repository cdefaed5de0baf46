#!/bin/bash
# dev tool (GPU box): tools/eval_locality.py under the kernel trace, both modes, for a view
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd /tmp; export TMPDIR=/tmp
for mode in band tiles; do
  D=/tmp/evloc_$1_$mode; rm -rf $D
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/tools/eval_locality.py $1 $mode > $D.txt 2>&1 || { echo FAILED; tail -3 $D.txt; exit 1; }
  grep "rows_used" $D.txt
  python3 - "$D" <<'PY'
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0])))
for r in rows:
    for k in ("march_defer", "eval_sample_rows", "composite_and_shade"):
        if k in r["Name"]: print(f'   {k:22s} calls {r["Calls"]:>3s}  avg {float(r["AverageNs"]) / 1e6:8.3f} ms  min {float(r["MinNs"]) / 1e6:8.3f}')
PY
done
