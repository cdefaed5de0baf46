#!/bin/bash
# dev tool (GPU box): PMC counters of the dominant kernel for one view.  usage: pmc_view.sh <view> <vol> COUNTER...
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
view=$1; vol=$2; shift; shift
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pmcv
timeout -k 10 150 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmcv -- python3 $R/tools/one_view.py $view $vol 2 > /tmp/pmcv.txt 2>&1
python3 - "$view" "$vol" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/pmcv/**/*counter_collection.csv", recursive=True)[0]
acc = {}
for r in csv.DictReader(open(f)):
    if int(r["Grid_Size"]) < 1000000 or "raymarch_pixels" not in r["Kernel_Name"]: continue
    acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k in sorted(acc): print(f"{sys.argv[1]:10s} vol={sys.argv[2]} {k:28s} {sum(acc[k])/len(acc[k]):.6g}")
PY
