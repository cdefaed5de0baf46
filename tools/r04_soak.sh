#!/bin/bash
# round 4 soaks on the final build: the randomized sweep (every launch variant incl. rounds / chains / tile maps) on three seeds,
# random scenes against the reference kernel live, both with many more cases than the suite's defaults
cd "$GRAFT_REPO_ROOT"
for seed in 41 42 43; do
  RRT_SWEEP_CASES=400 RRT_SWEEP_SEED=$seed timeout -k 10 900 python -m pytest tests/test_gpu_frames.py -m gpu -x -q -s -k "randomized_sweep" 2>&1 | grep -E "passed|failed|sweep:|Error" | tail -6
done
RRT_REF_SWEEP_CASES=150 timeout -k 10 600 python -m pytest tests/test_gpu_frames.py -m gpu -x -q -k "random_scenes_against_the_reference" 2>&1 | tail -2
