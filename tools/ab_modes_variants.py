"""dev tool (GPU): tools/ab_modes.py (4K bench frame, three arithmetic modes, byte hashes) for every library under
lib/variants, each in its own process via RRT_LIB_OVERRIDE, two rounds (same box, interleaved)."""
import glob, os, subprocess, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
libs = sorted(glob.glob(os.path.join(R, "relativisticraytracer_amd", "lib", "variants", "*.so")))
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    for lib in libs:
        r = subprocess.run([sys.executable, os.path.join(R, "tools", "ab_modes.py")], env=dict(os.environ, RRT_LIB_OVERRIDE=lib),
                           capture_output=True, text=True, timeout=600)
        for line in (r.stdout.strip() or r.stderr.strip()[-400:]).split("\n"):
            print(f"round {rnd} {os.path.basename(lib):20s} {line}", flush=True)
