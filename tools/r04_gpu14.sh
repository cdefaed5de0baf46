#!/bin/bash
cd "$GRAFT_REPO_ROOT"
V=relativisticraytracer_amd/lib/variants
for rep in 1 2; do
for lib in "" $V/maxrun16.so $V/maxrun8.so $V/maxrun4.so; do
  for view in default skimmer; do
    echo "== lib=${lib:-shipped(32)} view=$view"
    RRT_LIB_OVERRIDE=$lib python tools/shard_one.py 0 8 8 $view 2>&1 | grep "^frame" | tail -3 | cut -c1-175
  done
done
done
