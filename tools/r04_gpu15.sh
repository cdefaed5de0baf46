#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_t8.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/r04_t8.log
[ $rc -ne 0 ] && exit $rc
bash tools/r04_final_set.sh r04_c
