#!/bin/bash
# dev tool (GPU box): the three passes' kernel times (rocprofv3 kernel trace) of a full 4K view through the three-pass path, one chain,
# for every lib/variants/*.so.   usage: eval_time_variants.sh <view>
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd /tmp; export TMPDIR=/tmp
for f in $R/relativisticraytracer_amd/lib/variants/*.so; do
  D=/tmp/evt_$(basename $f .so); rm -rf $D
  RRT_LIB_OVERRIDE=$f RRT_CHAINS=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/tools/pass_workload.py $1 three_pass 4 > $D.txt 2>&1 || { echo FAILED $f; tail -3 $D.txt; exit 1; }
  echo "== $(basename $f)  $(grep rows_used $D.txt | cut -c1-60)"
  python3 - "$D" <<'PY'
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0])))
for r in rows:
    for k in ("march_defer", "eval_sample_rows", "composite_and_shade"):
        if k in r["Name"]: print(f'   {k:22s} calls {r["Calls"]:>3s}  avg {float(r["AverageNs"]) / 1e6:8.3f} ms  min {float(r["MinNs"]) / 1e6:8.3f}')
PY
done
