"""dev tool (GPU): rrt_selfcheck_div_march on 2^N march-shaped operand sets (default 2^40) -- the march's divides with the
reciprocal-root seeds the march itself produces (seeded Goldschmidt roots started from estimates off by up to their
acceptance tolerance) against IEEE `/`.  -> profiles/r04_div_march_seeds_probe.txt"""
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from relativisticraytracer_amd import _lib
lib = _lib.load_test()
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 40
chunk = 1 << 36
for tol1, tol2 in ((1.45e-4, 8.9e-3), (1.0e-4, 6.0e-3), (7.0e-5, 4.5e-3), (3.0e-5, 2.0e-3), (1.0e-5, 5.0e-4)):
    done = 0
    tot = [0] * 4
    last = None
    t0 = time.time()
    k = 0
    while done < (1 << logn):
        n = min(chunk, (1 << logn) - done)
        cnt = torch.zeros(8, dtype=torch.int64, device="cuda")
        _lib.check(lib.rrt_selfcheck_div_march(n, 4242 + k, tol1, tol2, C.c_void_p(cnt.data_ptr()), None), "selfcheck_div_march")
        torch.cuda.synchronize()
        for j in range(4):
            tot[j] += int(cnt[j])
        if int(cnt[0]) + int(cnt[1]):
            last = (int(cnt[4]), int(cnt[5]))
        done += n; k += 1
    print(f"seed errors uniform in +-{tol1:g} (one iteration) / +-{tol2:g} (two): 2^{logn} operand sets, accepted roots that are NOT the correctly "
          f"rounded one: {tot[0]} one-iteration, {tot[1]} two-iteration ({(tot[0] + tot[1]) / (1 << logn):.3g} per root); divides after a correct "
          f"root: {tot[2]} mismatches in {tot[3]} ({tot[2] / max(tot[3], 1):.3g} per divide); {time.time() - t0:.1f} s"
          + (f"; last failing root: x bits {last[0]:#010x} seed bits {last[1]:#010x}" if last else ""), flush=True)
