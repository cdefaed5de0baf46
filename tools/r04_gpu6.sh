#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r04_shard_skimmer -o shard -- python3 tools/shard_one.py 0 8 10 skimmer static > gpurun_out/r04_shard_one_skimmer.txt 2>&1
grep "^frame" gpurun_out/r04_shard_one_skimmer.txt | tail -3
head -9 gpurun_out/prof_r04_shard_skimmer/shard_kernel_stats.csv | cut -c1-170
