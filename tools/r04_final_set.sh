#!/bin/bash
# round 4 measurement set on ONE box: bench, rocprof trace + PMC passes, views, first-frame order, 8-shard projections, config 5
tag=${1:-r04_b}
cd "$GRAFT_REPO_ROOT"
rocm-smi --showclocks > gpurun_out/${tag}_rocm_smi.txt 2>&1
timeout -k 10 600 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
bash tools/profile_round.sh $tag > gpurun_out/${tag}_profile.log 2>&1; echo "profile rc=$?"
timeout -k 10 300 python tools/view_times.py > gpurun_out/${tag}_view_times.txt 2>&1; echo "views rc=$?"
timeout -k 10 300 python tools/first_frame.py > gpurun_out/${tag}_first_frame_order.txt 2>&1; echo "first rc=$?"
for v in default key1 skimmer; do timeout -k 10 300 python tools/shard_maps.py $v 8 2048 > gpurun_out/${tag}_shard_kernel_times_$v.txt 2>&1; echo "$v rc=$?"; done
timeout -k 10 300 python tools/shard_maps.py default 8 2048 0.99 > gpurun_out/${tag}_shard_kernel_times_config3_a099.txt 2>&1; echo "a099 rc=$?"
timeout -k 10 200 python tools/shard_scaling_probe.py > gpurun_out/${tag}_shard_scaling_probe.txt 2>&1; echo "scaling rc=$?"
timeout -k 10 300 relativisticraytracer_amd/lib/rrt_headless --width 7680 --height 4320 --frames 300 --path 0 --spin 0.9 --all-effects > gpurun_out/${tag}_headless_8k_path0_300frames.json 2> gpurun_out/${tag}_headless_8k.err; echo "config5 rc=$?"
cat gpurun_out/${tag}_headless_8k_path0_300frames.json
python - <<PY
import json
d=json.load(open("gpurun_out/${tag}_bench.json"))
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["clock_ghz"], d["roofline"]["frac_at_held_clock"], d["cpu_baseline"]["value"], d["fast_mode"]["ms_per_step"], {k:v["ms_per_step"] for k,v in d["heavy_view"].items() if isinstance(v,dict) and "ms_per_step" in v})
PY
