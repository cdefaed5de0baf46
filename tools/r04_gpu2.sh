#!/bin/bash
# round 4, second GPU call: new unit + frame tests, then the whole GPU suite, then the probe fit
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_units.py -m gpu -x -q -k "rk4 or divide" > gpurun_out/r04_t2_units.log 2>&1
echo "units rc=$?"; tail -3 gpurun_out/r04_t2_units.log
timeout -k 10 900 python -m pytest tests/test_gpu_frames.py -m gpu -x -q -k "rounds or tile_maps or first_frame or clock_probe or capture or cost_ordered or three_pass" --durations=10 > gpurun_out/r04_t2_frames.log 2>&1
rc=$?; echo "frames rc=$rc"; tail -15 gpurun_out/r04_t2_frames.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/probe_fit.py > gpurun_out/r04_probe_cost_fit.txt 2>&1
echo "fit rc=$?"; cat gpurun_out/r04_probe_cost_fit.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r04_t2_all.log 2>&1
echo "all rc=$?"; tail -25 gpurun_out/r04_t2_all.log
