"""dev tool (GPU): rrt_selfcheck_div on 2^34 march-shaped operand sets for the library named by RRT_LIB_OVERRIDE
(default: the shipped one) -> mismatches of the march's seeded divides against IEEE `/`.  Used to decide how many
Markstein corrections div_seeded needs (variants: -DRRT_DIV_ROUNDS=1 / 2; profiles/r03_div_rounds_probe.txt)."""
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from relativisticraytracer_amd import _lib
lib = _lib.load_test()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 34
cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
t0 = time.time()
_lib.check(lib.rrt_selfcheck_div(n, 777, C.c_void_p(cnt.data_ptr()), None), "selfcheck_div")
torch.cuda.synchronize()
print(f"{os.path.basename(os.environ.get('RRT_LIB_OVERRIDE', 'librrt_hip.so'))}: {n} operand sets (2 divides each), "
      f"{int(cnt[0])} mismatches ({int(cnt[0]) / (2 * n):.3g} per divide), {time.time() - t0:.1f} s; "
      f"last failing case: numerator bits {int(cnt[1]):#010x} denominator bits {int(cnt[2]):#010x} which {int(cnt[3])}")
