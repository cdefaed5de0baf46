#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_frames.py -m gpu -x -q -k "rounds or chains or three_pass or first_frame or cost_ordered or tile_maps or capture or sweep or config3" > gpurun_out/r04_t4.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -12 gpurun_out/r04_t4.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/first_frame.py > gpurun_out/r04_first_frame_order.txt 2>&1; echo "first rc=$?"; grep -v amdgpu.ids gpurun_out/r04_first_frame_order.txt
for v in default skimmer; do timeout -k 10 300 python tools/shard_maps.py $v 8 2048 > gpurun_out/r04_b_shard_maps_$v.txt 2>&1; echo "$v rc=$?"; grep -v amdgpu.ids gpurun_out/r04_b_shard_maps_$v.txt; done
