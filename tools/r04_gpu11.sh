#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for c in 1 0; do echo "== chains $c"; for v in default skimmer; do RRT_CHAINS=$c python tools/shard_one.py 0 8 8 $v 2>&1 | grep "^frame" | tail -2 | cut -c1-175; done; done
timeout -k 10 900 python -m pytest tests/test_gpu_frames.py -m gpu -x -q -k "rounds or chains or three_pass or sweep or config3 or tile_maps" > gpurun_out/r04_t5.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r04_t5.log
