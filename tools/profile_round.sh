#!/bin/bash
# dev tool (GPU box): rocprofv3 kernel trace + PMC passes of the bench command -> gpurun_out/prof_<tag>/
# usage: tools/profile_round.sh <tag> [bench args...]
tag=$1; shift
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/prof_$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 5 --warmup 1 --no-heavy --cpu-stride 0 "$@" > $OUT/trace_stdout.txt 2>&1
timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-stride 0 --no-fast --no-heavy "$@" > $OUT/pmc_fetch_stdout.txt 2>&1
timeout -k 10 280 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-stride 0 --no-fast --no-heavy "$@" > $OUT/pmc_write_stdout.txt 2>&1
timeout -k 10 280 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-stride 0 --no-fast --no-heavy "$@" > $OUT/pmc_sq_stdout.txt 2>&1
find $OUT -name "*.csv" | head -20
