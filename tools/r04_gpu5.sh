#!/bin/bash
cd "$GRAFT_REPO_ROOT"
V=relativisticraytracer_amd/lib/variants
for rep in 1 2; do
for lib in "" $V/defer7.so $V/defer6.so; do
  for view in default skimmer; do
    echo "== lib=${lib:-shipped} view=$view"
    RRT_LIB_OVERRIDE=$lib python tools/shard_one.py 0 8 8 $view 2>&1 | grep "^frame" | tail -4 | cut -c1-60
  done
done
done
