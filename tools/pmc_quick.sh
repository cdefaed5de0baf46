#!/bin/bash
# dev tool (GPU box): PMC counters per kernel for a quick_time.py run.  usage: pmc_quick.sh "<quick_time args>" COUNTER...
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
args=$1; shift
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pmcq
rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmcq -- python3 $R/tools/quick_time.py $args > /tmp/pmcq.txt 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/pmcq/**/*counter_collection.csv", recursive=True)[0]
acc = {}
for r in csv.DictReader(open(f)):
    k = (r["Kernel_Name"].split("(")[1].split("::")[-1][:40] if "anonymous" in r["Kernel_Name"] else r["Kernel_Name"][:30], r["Counter_Name"], int(r["Grid_Size"]))
    acc.setdefault(k, []).append(float(r["Counter_Value"]))
for k in sorted(acc):
    if k[2] > 1000000: print(f"{k[0]:42s} {k[1]:24s} grid {k[2]:9d}  avg {sum(acc[k])/len(acc[k]):.6g}")
PY
