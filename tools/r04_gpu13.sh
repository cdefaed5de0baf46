#!/bin/bash
cd "$GRAFT_REPO_ROOT"
RRT_DENSE_REF=1 RRT_DENSE_SHARDS=8 timeout -k 10 500 python tools/dense_parity.py 3840 2160 1 default > gpurun_out/r04_dense1.log 2>&1; echo "dense1 rc=$?"; grep -v amdgpu.ids gpurun_out/r04_dense1.log
RRT_DENSE_REF=1 RRT_DENSE_SHARDS=8 timeout -k 10 500 python tools/dense_parity.py 3840 2160 3 key1 grazing skimmer > gpurun_out/r04_dense2.log 2>&1; echo "dense2 rc=$?"; grep -v amdgpu.ids gpurun_out/r04_dense2.log
