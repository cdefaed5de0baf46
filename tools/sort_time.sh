#!/bin/bash
# dev tool (GPU box): the tile sort's kernels under the profiler (kernel trace of 'skimmer single_ordered') -> average us per launch
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
D=/tmp/sort_time; rm -rf $D; mkdir -p $D
cd /tmp; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python3 $R/tools/pass_workload.py skimmer single_ordered 6 > $D/trace.txt 2>&1 || exit 1
python3 - "$D" <<'PY'
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)[0])))
for r in rows:
    n = r["Name"]
    if "rrt_sort" in n or "probe" in n:
        print(f'{n.split("(")[0][-40:]:42s} calls {r["Calls"]:>4s}  avg {float(r["AverageNs"]) / 1e3:8.2f} us  min {float(r["MinNs"]) / 1e3:8.2f}  max {float(r["MaxNs"]) / 1e3:8.2f}')
PY
