#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for v in default key1 skimmer; do timeout -k 10 300 python tools/shard_maps.py $v 8 2048 > gpurun_out/r04_shard_kernel_times_$v.txt 2>&1; echo "$v rc=$?"; grep -v amdgpu.ids gpurun_out/r04_shard_kernel_times_$v.txt; done
timeout -k 10 300 python tools/shard_maps.py default 8 2048 0.99 > gpurun_out/r04_shard_kernel_times_config3_a099.txt 2>&1; echo "a099 rc=$?"; grep -v amdgpu.ids gpurun_out/r04_shard_kernel_times_config3_a099.txt
