#!/bin/bash
# dev tool (GPU box): the evidence set of a round on the FINAL sources, one call: rocprofv3 kernel trace + PMC passes of the bench command
# (tools/profile_round.sh), per-kernel counters of five (view, path) workloads (tools/pass_counters.sh), then the bench line itself.
# usage: tools/final_set.sh <tag>      -> gpurun_out/prof_<tag>/, gpurun_out/passc_*.json, gpurun_out/<tag>_bench.json
# (then, in the container: tools/summarize_profile.py <tag>; tools/summarize_pass_counters.py <tag>)
tag=$1
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
bash $R/tools/profile_round.sh $tag > $R/gpurun_out/${tag}_profile_round.log 2>&1 || exit 1
echo "profile_round done"
for vp in "default single" "key1 three_pass" "skimmer single" "skimmer single_ordered" "skimmer shard0of8"; do
  bash $R/tools/pass_counters.sh $vp > $R/gpurun_out/${tag}_passc_$(echo $vp | tr ' ' '_').log 2>&1 || { echo "pass_counters $vp FAILED"; exit 1; }
  echo "pass_counters $vp done"
done
cd $R && timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench done: $(cut -c1-120 gpurun_out/${tag}_bench.json)"
