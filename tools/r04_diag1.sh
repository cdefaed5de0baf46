#!/bin/bash
# round 4, first GPU call: cold / warm bring-up of the one-rank RCCL exchange (the r03_t6 hang), clock counters, headless tests
cd "$GRAFT_REPO_ROOT"
L=relativisticraytracer_amd/lib
export RRT_HEADLESS_TRACE=1
A="--width 160 --height 90 --frames 5 --path 0 --spin 0.9 --all-effects --init-timeout 400 --frame-timeout 100"
{ time RRT_NO_LIBRARY_WARMUP=1 NCCL_DEBUG=INFO $L/rrt_headless $A --force-collective ; } > gpurun_out/r04_hang_diag_cold.txt 2>&1
echo "cold rc=$?"
{ time $L/rrt_headless $A --force-collective ; } > gpurun_out/r04_hang_diag_warm.txt 2>&1
echo "warm rc=$?"
{ time $L/rrt_headless $A ; } > gpurun_out/r04_hang_diag_plain.txt 2>&1
echo "plain rc=$?"
ip addr > gpurun_out/r04_ip_addr.txt 2>&1 || cat /proc/net/dev > gpurun_out/r04_ip_addr.txt
tools/clock_probe > gpurun_out/r04_clock_probe.txt 2>&1
echo "clock rc=$?"
python -m pytest tests/test_headless.py -m gpu -x -q --durations=12 > gpurun_out/r04_t1.log 2>&1
echo "pytest rc=$?"
tail -5 gpurun_out/r04_t1.log
