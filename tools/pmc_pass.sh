#!/bin/bash
# dev tool (GPU box): one rocprofv3 --pmc pass over the strict bench kernel; prints per-launch averages
# usage: tools/pmc_pass.sh <outname> COUNTER [COUNTER...]
name=$1; shift
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/pmc_$name
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc "$@" --output-format csv -d $OUT -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-stride 0 --no-fast > $OUT/stdout.txt 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc, n = {}, {}
for r in csv.DictReader(open(f)):
    if "raymarch" not in r["Kernel_Name"]: continue
    k = r["Counter_Name"]; acc[k] = acc.get(k, 0) + float(r["Counter_Value"]); n[k] = n.get(k, 0) + 1
for k in sorted(acc): print(f"{k:32s} {acc[k]/n[k]:.6g}")
PY
