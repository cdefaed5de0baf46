"""dev tool (GPU): the memset-node question under TORCH's capture (torch.cuda.graph), which is how the failing library test captures.
A hipMalloc'ed buffer (not torch memory), hipMemsetAsync through ctypes on torch's capture stream, torch ops as reader / dirtier."""
import ctypes as C
import torch
hip = C.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
torch.cuda.init(); torch.zeros(1, device="cuda")
bad_total = 0
for nbytes in (1792, 4352, 259584):
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), 64 << 20) == 0
    n = nbytes // 4
    # a torch view of the foreign allocation
    class _Arr:      # __cuda_array_interface__ wrapper
        def __init__(self, ptr, n): self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i4", "data": (ptr, False), "version": 2}
    buf = torch.as_tensor(_Arr(p.value, n), device="cuda")
    buf.fill_(7)
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        st = torch.cuda.current_stream().cuda_stream
        assert hip.hipMemsetAsync(p, 0, nbytes, C.c_void_p(st)) == 0
        out.copy_(buf)
        buf.add_(7)
    for rep in range(4):
        out.fill_(-1)
        g.replay()
        torch.cuda.synchronize()
        stale = (out != 0).nonzero().flatten()
        bad = int(stale.numel())
        bad_total += bad
        where = ""
        if bad:
            contiguous = bool((stale[1:] - stale[:-1] == 1).all())
            where = f"   <-- STALE: words {int(stale[0])}..{int(stale[-1])}{' (contiguous)' if contiguous else ' (not contiguous)'}, value {int(out[stale[0]])}"
        print(f"torch capture, {nbytes} bytes, replay {rep}: {bad} of {n} words not cleared" + where)
print("MEMSET NODE DEFECT REPRODUCED under torch.cuda.graph" if bad_total else "clean under torch.cuda.graph as well")
