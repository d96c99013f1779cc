"""tools/exp_ctx_placement.py — does the speed of the scan depend on WHERE a context's workspace landed?  Several contexts of one
library in one process, created one after the other (with dummy allocations of different sizes in between), all scanning the
same 16 GiB buffer."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
n = (16 << 30) // 330 * 330
buf = torch.empty(n + 4096, dtype=torch.uint8, device=dev)
cap = n // 300 + 16
rs = torch.empty(cap, dtype=torch.int64, device=dev)
L = C.CDLL(os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else "fastq-rs_amd/libfastq_hip.so"))
L.fqh_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
L.fqh_synth_fill.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
L.fqh_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
L.fqh_last_timing.argtypes = [C.c_void_p, C.c_void_p]
summ = (C.c_uint64 * 16)(); carry = (C.c_uint64 * 8)()
ctxs = []; pads = []
for i, pad_mb in enumerate([0, 0, 1, 37, 512, 3000, 0, 0]):
    if pad_mb:
        pads.append(torch.empty(pad_mb << 20, dtype=torch.uint8, device=dev))
    h = C.c_void_p()
    assert L.fqh_create(0, C.byref(h)) == 0
    if i == 0:
        assert L.fqh_synth_fill(h, buf.data_ptr(), 0, n, 0x5EEDF00D2026) == 0
    assert L.fqh_scan(h, buf.data_ptr(), n, 1, None, rs.data_ptr(), cap, summ, carry) == 0   # allocates the workspace now
    ctxs.append((i, pad_mb, h))
for rnd in range(3):
    for i, pad_mb, h in ctxs:
        for _ in range(3): L.fqh_scan(h, buf.data_ptr(), n, 1, None, rs.data_ptr(), cap, summ, carry)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): L.fqh_scan(h, buf.data_ptr(), n, 1, None, rs.data_ptr(), cap, summ, carry)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20 * 1e3
        t = (C.c_float * 5)(); L.fqh_last_timing(h, t)
        print("round %d context %d (pad %4d MB before it): %.3f ms per step (index %.3f emit %.3f)" % (rnd, i, pad_mb, dt, t[1], t[3]), flush=True)
