#!/bin/bash
# round 4 measurement set "a": tests, bench, profile (trace + PMC), views, config 5, divide probe
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=6 > gpurun_out/r04_t7.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -12 gpurun_out/r04_t7.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python bench.py > gpurun_out/r04_a_bench.json 2> gpurun_out/r04_a_bench.err; echo "bench rc=$?"
bash tools/profile_round.sh r04_a > gpurun_out/r04_a_profile.log 2>&1; echo "profile rc=$?"
timeout -k 10 300 python tools/view_times.py > gpurun_out/r04_a_view_times.txt 2>&1; echo "views rc=$?"
timeout -k 10 300 relativisticraytracer_amd/lib/rrt_headless --width 7680 --height 4320 --frames 300 --path 0 --spin 0.9 --all-effects > gpurun_out/r04_a_headless_8k_path0_300frames.json 2> gpurun_out/r04_a_headless_8k.err; echo "config5 rc=$?"; cat gpurun_out/r04_a_headless_8k_path0_300frames.json
timeout -k 10 400 python tools/div_march_probe.py 41 > gpurun_out/r04_div_march_seeds_probe.txt 2>&1; echo "div rc=$?"; tail -2 gpurun_out/r04_div_march_seeds_probe.txt
