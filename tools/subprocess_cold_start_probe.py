"""dev tool (GPU): how long does the FIRST child process that uses the GPU take while this process holds a context?
(tests/test_headless.py spawns the drivers from inside pytest; on some boxes the first such child is slow.)"""
import os
import subprocess
import sys
import time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
t0 = time.perf_counter()
import torch
x = torch.zeros(1 << 20, device="cuda"); torch.cuda.synchronize()
print(f"parent: torch + context in {time.perf_counter() - t0:.2f} s", flush=True)
env = dict(os.environ, RRT_HEADLESS_TRACE="1")
for i in range(3):
    t = time.perf_counter()
    r = subprocess.run([sys.executable, "-m", "relativisticraytracer_amd.headless", "--width", "96", "--height", "54", "--frames", "3",
                        "--path", "0", "--spin", "0.9", "--all-effects", "--out", "/tmp/probe.rgba"], cwd=R, env=env, capture_output=True, text=True)
    print(f"child {i} (python driver): {time.perf_counter() - t:.2f} s rc={r.returncode}\n{r.stderr.strip()}", flush=True)
exe = os.path.join(R, "relativisticraytracer_amd", "lib", "rrt_headless")
for i in range(2):
    t = time.perf_counter()
    r = subprocess.run([exe, "--width", "96", "--height", "54", "--frames", "3", "--path", "0", "--spin", "0.9", "--out", "/tmp/probe2.rgba"],
                       env=env, capture_output=True, text=True)
    print(f"child {i} (C++ driver): {time.perf_counter() - t:.2f} s rc={r.returncode}\n{r.stderr.strip()[-600:]}", flush=True)
