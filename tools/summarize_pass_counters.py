"""dev tool (container): condense gpurun_out/passc_<view>_<path>.json (tools/pass_counters.sh) into profiles/<tag>_passc_*.json and
profiles/pass_counters.json -- the per-kernel counter record bench.py quotes (heavy_view / three-pass kernels), keyed to the
hash of the kernel sources it was measured on.      usage: summarize_pass_counters.py <tag>"""
import glob, json, os, shutil, sys
sys.path.insert(0, ".")
import bench
tag = sys.argv[1]
rec = {"source_hash": bench.source_hash(), "from": f"profiles/{tag}_passc_*.json (tools/pass_counters.sh: kernel trace + separate PMC passes)",
       "note": "per 4K frame; issue_slot_util = SQ_INSTS_VALU x 2 clocks / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); hbm_bytes = "
               "(2 FETCH_SIZE + WRITE_SIZE) KB, an upper bound (FETCH_SIZE doubled as the microarchitecture guide prescribes for gfx950)",
       "runs": {}}
for f in sorted(glob.glob("gpurun_out/passc_*.json")):
    d = json.load(open(f))
    shutil.copy(f, f"profiles/{tag}_{os.path.basename(f)}")
    key = f"{d['view']}/{d['path']}"
    ks = {}
    for k, v in d["kernels"].items():
        if v.get("total_ms", 0) / max(d.get("frames_traced", 4), 1) < 0.01 and k not in ("scatter", "histogram"):
            continue
        ks[k] = {"ms_per_frame": round(v.get("total_ms", 0) / d.get("frames_traced", 4), 4),
                 "launches_per_frame": round(v.get("calls", 0) / d.get("frames_traced", 4), 2),
                 "valu_insts": v.get("SQ_INSTS_VALU_per_frame"), "issue_slot_util": round(v["issue_slot_util"], 4) if "issue_slot_util" in v else None,
                 "hbm_bytes_upper": v.get("hbm_bytes_per_frame_upper"),
                 "fetch_kb": v.get("FETCH_SIZE_per_frame"), "write_kb": v.get("WRITE_SIZE_per_frame")}
    rec["runs"][key] = ks
json.dump(rec, open("profiles/pass_counters.json", "w"), indent=1)
print(json.dumps(rec, indent=1)[:3000])
