#!/bin/bash
# dev tool (GPU box): per-KERNEL durations and counters of one view through one path (tools/pass_workload.py):
#   kernel trace + stats, then separate PMC passes (SQ set, FETCH_SIZE, WRITE_SIZE) -> gpurun_out/passc_<view>_<path>.json
# usage: tools/pass_counters.sh <view> <path>        (the program after `--` is python3 itself: no env, no shell in between)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
view=$1; path=$2
D=/tmp/passc_${view}_${path}; rm -rf $D; mkdir -p $D $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python3 $R/tools/pass_workload.py $view $path 4 > $D/trace.txt 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $D/sq -- python3 $R/tools/pass_workload.py $view $path 2 > $D/sq.txt 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/fetch -- python3 $R/tools/pass_workload.py $view $path 2 > $D/fetch.txt 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/write -- python3 $R/tools/pass_workload.py $view $path 2 > $D/write.txt 2>&1 || exit 1
python3 - "$view" "$path" "$D" "$R/gpurun_out/passc_${view}_${path}.json" <<'PY'
import csv, glob, json, re, sys
view, path, D, out = sys.argv[1:5]
def short(n):
    m = re.search(r"(raymarch_pixels|march_defer|eval_sample_rows|composite_and_shade|pool_next_round|zero_words|probe_costs|probe_to_tiles|histogram|scan_counters|scatter|build_noise_table)", n)
    return m.group(1) if m else None
res = {"view": view, "path": path, "kernels": {}}
kt = glob.glob(f"{D}/trace/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(kt[0])))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
for r in rows:
    k = short(r["Kernel_Name"])
    if not k or k == "build_noise_table": continue
    d = res["kernels"].setdefault(k, {"calls": 0, "total_ms": 0.0})
    d["calls"] += 1; d["total_ms"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for k, d in res["kernels"].items():
    d["avg_ms"] = d["total_ms"] / d["calls"]
res["frames_traced"] = 4
for name in ("sq", "fetch", "write"):
    cs = glob.glob(f"{D}/{name}/**/*counter_collection.csv", recursive=True)
    if not cs: continue
    acc = {}
    for r in csv.DictReader(open(cs[0])):
        k = short(r["Kernel_Name"])
        if not k or k == "build_noise_table": continue
        acc.setdefault(k, {}).setdefault(r["Counter_Name"], 0.0)
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, c in acc.items():
        for cn, v in c.items():
            res["kernels"].setdefault(k, {})[cn + "_per_frame"] = v / 2.0          # two frames per PMC pass
for k, d in res["kernels"].items():
    if "FETCH_SIZE_per_frame" in d and "WRITE_SIZE_per_frame" in d:
        # KB; FETCH_SIZE doubled on gfx950 for wide reads (MI355X_MICROARCH.md) -- an upper bound
        d["hbm_bytes_per_frame_upper"] = (2 * d["FETCH_SIZE_per_frame"] + d["WRITE_SIZE_per_frame"]) * 1024
    if "SQ_INSTS_VALU_per_frame" in d and "GRBM_GUI_ACTIVE_per_frame" in d and d["GRBM_GUI_ACTIVE_per_frame"] > 0:
        # issue-slot utilisation: VALU instructions x 2 cycles / (1024 SIMDs x the clocks the kernel's launches were active);
        # GRBM_GUI_ACTIVE sums over the 8 XCDs
        d["issue_slot_util"] = d["SQ_INSTS_VALU_per_frame"] * 2.0 / (1024.0 * d["GRBM_GUI_ACTIVE_per_frame"] / 8.0)
res["stdout"] = open(f"{D}/trace.txt").read()[-600:]
json.dump(res, open(out, "w"), indent=1)
for k, d in sorted(res["kernels"].items(), key=lambda kv: -kv[1].get("total_ms", 0)):
    print(f"{view:8s} {path:14s} {k:22s} calls/frame {d.get('calls', 0) / 4:5.1f}  ms/frame {d.get('total_ms', 0) / 4:8.3f}  VALU {d.get('SQ_INSTS_VALU_per_frame', 0):.4g}  "
          f"util {d.get('issue_slot_util', 0):.3f}  fetchKB {d.get('FETCH_SIZE_per_frame', 0):.4g} writeKB {d.get('WRITE_SIZE_per_frame', 0):.4g}")
PY
