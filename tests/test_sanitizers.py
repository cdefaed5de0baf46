"""AddressSanitizer + UndefinedBehaviorSanitizer runs of the CPU-side code.
CPU only: GPU ASan / XNACK are not available on this pool.

  * oracle/rrt_oracle.c -- the restatement every parity claim rests on -- through tests/sanitize/oracle_exerciser.c:
    small frames in both math modes, rect / strided renders, the gate recorder and its replay, the unit functions on
    zeros, negative lattice points, huge coordinates, NaN and infinity;
  * the HOST side of librrt_hip.so (handle registries, noise-table planning and the drivers' window policy, camera
    basis / path playback / recording clock, argument checks, the device-binding checks through the fake-device
    hook) through tests/sanitize/host_exerciser.cpp, against a build of csrc/rrt_hip.hip with host-only
    instrumentation (hipcc -fsanitize=address,undefined -fno-gpu-sanitize).  No kernel is launched.
The sanitizers abort the process on a finding (-fno-sanitize-recover), so exit code 0 means clean.
"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", OMP_NUM_THREADS="4")


def _make(target):
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), target], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]


def test_oracle_restatement_is_clean_under_asan_and_ubsan():
    _make("asan")
    r = subprocess.run([os.path.join(ROOT, "oracle", "_asan", "oracle_exerciser")], capture_output=True, text=True, env=ENV, timeout=900)
    assert r.returncode == 0 and "oracle exerciser ok" in r.stdout, (r.stdout[-800:], r.stderr[-3000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]


def test_library_host_side_is_clean_under_asan_and_ubsan():
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not present: the host-instrumented library cannot be built here")
    _make("asan-host")
    r = subprocess.run([os.path.join(ROOT, "oracle", "_asan", "host_exerciser")], capture_output=True, text=True, env=ENV, timeout=900)
    assert r.returncode == 0 and "host exerciser ok" in r.stdout, (r.stdout[-800:], r.stderr[-3000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
