#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r04_t3_all.log 2>&1
rc=$?; echo "all rc=$rc"; tail -16 gpurun_out/r04_t3_all.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/first_frame.py > gpurun_out/r04_first_frame_order.txt 2>&1; echo "first rc=$?"; grep -v amdgpu.ids gpurun_out/r04_first_frame_order.txt
timeout -k 10 300 python tools/probe_fit.py > gpurun_out/r04_probe_cost_fit2.txt 2>&1; echo "fit rc=$?"; grep -v amdgpu.ids gpurun_out/r04_probe_cost_fit2.txt
timeout -k 10 600 python bench.py --steps 10 --warmup 2 > gpurun_out/r04_a_bench.json 2> gpurun_out/r04_a_bench.err; echo "bench rc=$?"; cut -c1-1500 gpurun_out/r04_a_bench.json
