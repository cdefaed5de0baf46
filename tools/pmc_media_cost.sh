#!/bin/bash
# dev tool (GPU box): PMC counters of tools/media_cost.py's unit-kernel dispatches: VALU instructions per sample (wave) against the time-derived issue slots
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pmcm
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD --output-format csv -d /tmp/pmcm -- python3 $R/tools/media_cost.py > /tmp/pmcm.txt 2>&1
grep -v amdgpu /tmp/pmcm.txt
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/pmcm/**/*counter_collection.csv", recursive=True)[0]
rows = {}
for r in csv.DictReader(open(f)):
    if "unit" not in r["Kernel_Name"] and "media" not in r["Kernel_Name"]: continue
    rows.setdefault(int(r["Dispatch_Id"]), {"k": r["Kernel_Name"][:60], "vgpr": r.get("VGPR_Count", r.get("Arch_VGPR_Count", "?"))})[r["Counter_Name"]] = float(r["Counter_Value"])
for d in sorted(rows):
    x = rows[d]
    if x.get("SQ_WAVES", 0) < 60000: continue
    w = x["SQ_WAVES"]; cyc = x["GRBM_GUI_ACTIVE"] / 8
    print(f"dispatch {d:4d} vgpr {x['vgpr']:>4} waves {w:8.0f}  VALU/wave {x['SQ_INSTS_VALU'] / w:8.1f}  VMEM_RD/wave {x.get('SQ_INSTS_VMEM_RD', 0) / w:6.1f}  "
          f"time-derived slots/wave {cyc * 1024 / 2 / w:8.1f}  issue efficiency {x['SQ_INSTS_VALU'] * 2 / (cyc * 1024):.3f}  avg waves/SIMD {x['SQ_WAVE_CYCLES'] * 4 / (cyc * 1024):.2f}")
PY
