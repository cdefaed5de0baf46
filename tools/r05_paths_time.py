"""dev tool (GPU): wall time per frame of tools/pass_workload.py's paths for several views, in ONE process (same box, same clocks).
usage: r05_paths_time.py [views] [paths]"""
import os, subprocess, sys
views = (sys.argv[1] if len(sys.argv) > 1 else "default,key1,grazing,skimmer").split(",")
paths = (sys.argv[2] if len(sys.argv) > 2 else "single,single_ordered,three_pass,shard0of8,shard0of8_ordered").split(",")
here = os.path.dirname(os.path.abspath(__file__))
for v in views:
    for p in paths:
        env = dict(os.environ)
        if p == "three_pass":
            env["RRT_CHAINS"] = "2"
        r = subprocess.run([sys.executable, os.path.join(here, "pass_workload.py"), v, p, "6"], capture_output=True, text=True, env=env)
        ms = [float(l.split(":")[1].split()[0]) for l in r.stdout.splitlines() if " ms" in l]
        tail = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(f"{v:8s} {p:20s} min {min(ms[2:]):8.3f}  median {sorted(ms[2:])[len(ms[2:]) // 2]:8.3f} ms   {tail[0] if tail else ''}", flush=True)
