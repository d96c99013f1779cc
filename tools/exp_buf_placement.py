"""tools/exp_buf_placement.py — does the speed of the scan depend on where the INPUT landed?  One context (one workspace), several
16 GiB inputs allocated one after the other in one process, the same bytes in each."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
n = (16 << 30) // 330 * 330
cap = n // 300 + 16
L = C.CDLL(os.path.abspath("fastq-rs_amd/libfastq_hip.so"))
L.fqh_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
L.fqh_synth_fill.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
L.fqh_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
L.fqh_last_timing.argtypes = [C.c_void_p, C.c_void_p]
L.fqh_read_ceiling.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
summ = (C.c_uint64 * 16)(); carry = (C.c_uint64 * 8)()
rs = torch.empty(cap, dtype=torch.int64, device=dev)
bufs = [torch.empty(n + 4096, dtype=torch.uint8, device=dev) for _ in range(6)]
ctxs = []
for j in range(2):
    h = C.c_void_p(); assert L.fqh_create(0, C.byref(h)) == 0
    ctxs.append(h)
for b in bufs:
    assert L.fqh_synth_fill(ctxs[0], b.data_ptr(), 0, n, 0x5EEDF00D2026) == 0
for rnd in range(2):
    for j, h in enumerate(ctxs):
        for i, b in enumerate(bufs):
            for _ in range(3): assert L.fqh_scan(h, b.data_ptr(), n, 1, None, rs.data_ptr(), cap, summ, carry) == 0
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(15): L.fqh_scan(h, b.data_ptr(), n, 1, None, rs.data_ptr(), cap, summ, carry)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 15 * 1e3
            t = (C.c_float * 5)(); L.fqh_last_timing(h, t)
            print("round %d context %d input %d (%x): %.3f ms per step (index %.3f emit %.3f)" % (rnd, j, i, b.data_ptr(), dt, t[1], t[3]), flush=True)
# the bare read of every input (k_read_ceiling): is it the input alone, or the input next to the scan's line stores?
cs = C.c_uint64(); ms = C.c_float()
for rnd in range(2):
    for i, b in enumerate(bufs):
        best = 1e9
        for _ in range(3):
            assert L.fqh_read_ceiling(ctxs[0], b.data_ptr(), n, C.byref(cs), C.byref(ms)) == 0
            best = min(best, ms.value)
        print("bare read, input %d: %.3f ms" % (i, best), flush=True)
