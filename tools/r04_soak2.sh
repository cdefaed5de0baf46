#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for seed in 2026 10042026; do
  RRT_SWEEP_CASES=1200 RRT_SWEEP_SEED=$seed timeout -k 10 1000 python -m pytest tests/test_gpu_frames.py -m gpu -x -q -s -k "randomized_sweep" 2>&1 | grep -E "passed|failed|sweep: (400|800|1200)|Error" | tee -a gpurun_out/r04_soaks2.txt
done
RRT_REF_SWEEP_CASES=400 RRT_SWEEP_SEED=77 timeout -k 10 900 python -m pytest tests/test_gpu_frames.py -m gpu -x -q -k "random_scenes_against_the_reference" 2>&1 | tail -2 | tee -a gpurun_out/r04_soaks2.txt
