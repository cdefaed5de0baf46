#!/bin/bash
# dev tool (GPU box): per-kernel sums of arbitrary PMC counter SETS for one view through one path (tools/pass_workload.py), one
# rocprofv3 pass per set (no trace options beside --pmc).  usage: tools/pmc_diag.sh <view> <path> "<SET A counters>" "<SET B counters>" ...
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
view=$1; path=$2; shift; shift
cd /tmp; export TMPDIR=/tmp
k=0
for set in "$@"; do
  D=/tmp/pmcd_${view}_${path}_$k; rm -rf $D
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $D -- python3 $R/tools/pass_workload.py $view $path 2 > $D.txt 2>&1 || { echo "FAILED set $k: $set"; tail -3 $D.txt; }
  python3 - "$view" "$path" "$D" <<'PY'
import csv, glob, re, sys
view, path, D = sys.argv[1:4]
cs = glob.glob(f"{D}/**/*counter_collection.csv", recursive=True)
acc = {}
for r in (csv.DictReader(open(cs[0])) if cs else []):
    m = re.search(r"(raymarch_pixels|march_defer|eval_sample_rows|composite_and_shade)", r["Kernel_Name"])
    if not m: continue
    acc.setdefault(m.group(1), {}).setdefault(r["Counter_Name"], 0.0)
    acc[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"]) / 2.0        # two frames
for k, c in acc.items():
    print(f"{view:8s} {path:12s} {k:20s} " + "  ".join(f"{n} {v:.4g}" for n, v in sorted(c.items())))
PY
  k=$((k+1))
done
