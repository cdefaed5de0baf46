#!/bin/bash
# dev tool (GPU box): kernel timeline of the last frame of tools/shard_one.py <shard> <N> 4 <view> (rocprofv3 --kernel-trace)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
view=${1:-default}; out=gpurun_out/prof_chain_$view
rocprofv3 --kernel-trace --output-format csv -d $out -o ch -- python3 tools/shard_one.py 0 8 4 $view > $out.txt 2>&1
grep "^frame" $out.txt | tail -2 | cut -c1-170
python - <<PY
import csv
rows=list(csv.DictReader(open("$out/ch_kernel_trace.csv")))
rows=[r for r in rows if any(k in r["Kernel_Name"] for k in ("march_defer","eval_sample","composite","pool_next"))]
last=[r for r in rows if "march_defer" in r["Kernel_Name"] and "false>(" in r["Kernel_Name"].split("FrameArgs")[0]]
# the last frame: from the last pair of RESUME=false marches on
starts=sorted(int(r["Start_Timestamp"]) for r in rows if "march_defer" in r["Kernel_Name"] and ", false>" in r["Kernel_Name"])
t0=starts[-2] if len(starts)>1 else starts[-1]
for r in rows:
    if int(r["Start_Timestamp"]) < t0: continue
    n=r["Kernel_Name"]; n=n.split("::")[-2 if n.count("::")>1 else -1] if False else n[n.find("namespace)::")+12:][:44]
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
    if d < 0.02: continue
    print(f'{(int(r["Start_Timestamp"])-t0)/1e6:8.3f} -> {(int(r["End_Timestamp"])-t0)/1e6:8.3f} ms ({d:6.3f})  q{r.get("Queue_Id","?")}  {n[:60]}')
PY
