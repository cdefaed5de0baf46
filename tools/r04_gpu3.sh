#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for v in default skimmer; do
  timeout -k 10 400 python tools/shard_maps.py $v 8 2048 > gpurun_out/r04_a_shard_maps_$v.txt 2>&1
  echo "$v rc=$?"; cat gpurun_out/r04_a_shard_maps_$v.txt
done
