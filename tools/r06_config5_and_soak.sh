#!/bin/bash
# round 6's one batch on the final sources (GPU box): box calibration (4K bench frame, three modes), config 5 through rrt_headless on one GPU
# (strict and FMAD), its 8-rank projection, one 600-scene randomized soak, dense parity of the 4K bench frame (every pixel) and three
# disk-heavy views (every 3rd) against the oracle and the reference kernel body, also as 8 shards.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); cd $R
O=gpurun_out
python tools/ab_modes.py > $O/r06_c_box_calibration.txt 2>&1; cat $O/r06_c_box_calibration.txt
H=relativisticraytracer_amd/lib/rrt_headless
timeout -k 10 200 $H --width 7680 --height 4320 --frames 300 --path 0 --spin 0.9 --all-effects > $O/r06_c_headless_8k_path0_300frames_strict.json 2> $O/r06_c_h8k_strict.err; cat $O/r06_c_headless_8k_path0_300frames_strict.json
timeout -k 10 200 $H --width 7680 --height 4320 --frames 300 --path 0 --spin 0.9 --all-effects --arith fmad > $O/r06_c_headless_8k_path0_300frames_fmad.json 2> $O/r06_c_h8k_fmad.err; cat $O/r06_c_headless_8k_path0_300frames_fmad.json
timeout -k 10 300 python tools/config5_projection.py 0 > $O/r06_c_config5_projection.txt 2>&1; tail -6 $O/r06_c_config5_projection.txt
timeout -k 10 300 python tools/config5_projection.py 2 > $O/r06_c_config5_projection_fmad.txt 2>&1; tail -2 $O/r06_c_config5_projection_fmad.txt
RRT_SWEEP_CASES=600 RRT_SWEEP_SEED=601 timeout -k 10 400 python -m pytest tests/test_gpu_frames.py -m gpu -q -k randomized_sweep > $O/r06_c_soak_601.txt 2>&1; tail -2 $O/r06_c_soak_601.txt
RRT_DENSE_REF=1 RRT_DENSE_SHARDS=8 timeout -k 10 600 python tools/dense_parity.py 3840 2160 1 default > $O/r06_c_dense_parity.txt 2>&1
RRT_DENSE_REF=1 RRT_DENSE_SHARDS=8 timeout -k 10 600 python tools/dense_parity.py 3840 2160 3 key1 grazing skimmer >> $O/r06_c_dense_parity.txt 2>&1; cat $O/r06_c_dense_parity.txt
