#!/bin/bash
# dev tool (GPU box): tools/sort_time.sh for the shipped library and every lib/variants/*.so
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
echo "== shipped"; bash $R/tools/sort_time.sh | grep rrt_sort
for f in $R/relativisticraytracer_amd/lib/variants/*.so; do echo "== $(basename $f)"; RRT_LIB_OVERRIDE=$f bash $R/tools/sort_time.sh | grep rrt_sort; done
