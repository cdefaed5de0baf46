#!/bin/bash
# dev tool (GPU box): kernel trace of tools/overlap_probe.py for shard 0 of 8; prints how much of the busy
# time has kernels of two different frames (streams) resident at once.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_ov
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ov -- python3 $R/tools/overlap_probe.py 8 > /tmp/prof_ov.txt 2>&1
grep "^N=" /tmp/prof_ov.txt
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/prof_ov/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in ("march_defer", "eval_sample", "composite"))]
ev = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))
# last 24 frames (72 kernels) = the second "two streams" repetition; the 72 before them = the "one stream" one...
def overlap(rs):
    pts = []
    for r in rs:
        pts.append((int(r["Start_Timestamp"]), 1)); pts.append((int(r["End_Timestamp"]), -1))
    pts.sort()
    busy = both = 0; depth = 0; last = pts[0][0]
    for t, d in pts:
        if depth >= 1: busy += t - last
        if depth >= 2: both += t - last
        depth += d; last = t
    return busy / 1e6, both / 1e6
k = 72
groups = [ev[i:i + k] for i in range(0, len(ev), k)]
names = ["one stream (rep 1)", "one stream (rep 2)", "two streams (rep 1)", "two streams (rep 2)"]
for name, g in zip(names, groups):
    busy, both = overlap(g)
    span = (max(int(r["End_Timestamp"]) for r in g) - min(int(r["Start_Timestamp"]) for r in g)) / 1e6
    queues = sorted({r.get("Queue_Id", "?") for r in g})
    print(f"{name:22s}: {len(g)} kernels on queues {queues}: span {span:.2f} ms ({span / 24:.3f} ms/frame), >=2 kernels resident {both:.2f} ms ({100 * both / busy:.0f} % of busy time)")
PY
