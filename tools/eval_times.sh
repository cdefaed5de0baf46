#!/bin/bash
# dev tool (GPU box): kernel times of the three-pass path for one view.  usage: eval_times.sh <view> <table> [w h gib]
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/evt
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/evt -- python3 $R/tools/eval_times.py "$@" > /tmp/evt.txt 2>&1
tail -1 /tmp/evt.txt
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/evt/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if any(k in n for k in ("march_defer", "eval_sample", "composite", "raymarch_pixels")):
        print(f"   {n.split('(')[0][-60:]:60s} calls {r['Calls']:>3s} avg {float(r['AverageNs'])/1e6:8.3f} ms")
PY
