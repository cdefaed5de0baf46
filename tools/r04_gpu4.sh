#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for mode in static order; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r04_shard_$mode -o shard -- python3 tools/shard_one.py 0 8 10 default $mode > gpurun_out/r04_shard_one_$mode.txt 2>&1
echo "rc=$?"; tail -4 gpurun_out/r04_shard_one_$mode.txt
f=$(find gpurun_out/prof_r04_shard_$mode -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-200
done
