#!/bin/bash
# dev tool (GPU box): per-kernel times (rocprofv3 --stats) of a python tool run.  usage: time_passes.sh <script> [args...]
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
script=$1; shift
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_tp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tp -- python3 $R/tools/$script "$@" > /tmp/prof_tp.txt 2>&1
grep -v "amdgpu\|rocprofv3\|^W2026\|^E2026" /tmp/prof_tp.txt | tail -4
cat /tmp/prof_tp/*/*kernel_stats.csv | grep -E "march_defer|eval_sample|composite|raymarch_pixels" | python3 -c "import csv,sys; [print('   %-60s calls %s avg %.3f ms min %.3f max %.3f' % (r[0][:60], r[1], float(r[3])/1e6, float(r[5])/1e6, float(r[6])/1e6)) for r in csv.reader(sys.stdin)]"
