#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for c in 1 2; do
  echo "== chains $c"; RRT_CHAINS=$c python tools/shard_one.py 0 8 8 default 2>&1 | grep "^frame" | tail -3 | cut -c1-150
done
RRT_CHAINS=2 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_r04_chains -o ch -- python3 tools/shard_one.py 0 8 4 default > /dev/null 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_r04_chains/ch_kernel_trace.csv")))
rows=[r for r in rows if any(k in r["Kernel_Name"] for k in ("march_defer","eval_sample","composite","pool_next"))]
t0=min(int(r["Start_Timestamp"]) for r in rows[-40:])
for r in rows[-24:]:
    n=r["Kernel_Name"]; n=n[n.find("::")+2:n.find("(")][:44]
    print(f'{(int(r["Start_Timestamp"])-t0)/1e6:8.3f} -> {(int(r["End_Timestamp"])-t0)/1e6:8.3f} ms  q{r.get("Queue_Id","?")}  grid {r.get("Grid_Size_X","?")}x{r.get("Grid_Size_Y","?")}  {n}')
PY
