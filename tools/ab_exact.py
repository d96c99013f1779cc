"""tools/ab_exact.py LIB [LIB ...] — the exact path's scan step (FQH_OPT_FAST_PATH 0: k_index_t + k_emit) over 16 GiB, builds interleaved."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
n = (16 << 30) // 330 * 330
buf = torch.empty(n + 4096, dtype=torch.uint8, device=dev)
cap = n // 300 + 16
rs = torch.empty(cap, dtype=torch.int64, device=dev)
libs = []
for path in sys.argv[1:]:
    L = C.CDLL(os.path.abspath(path))
    h = C.c_void_p()
    L.fqh_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    assert L.fqh_create(0, C.byref(h)) == 0
    L.fqh_synth_fill.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
    L.fqh_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    L.fqh_set_option.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.fqh_last_timing.argtypes = [C.c_void_p, C.c_void_p]
    assert L.fqh_set_option(h, 1, 0) == 0
    libs.append((os.path.basename(path), L, h))
assert libs[0][1].fqh_synth_fill(libs[0][2], buf.data_ptr(), 0, n, 0x5EEDF00D2026) == 0
summ = (C.c_uint64 * 16)(); carry = (C.c_uint64 * 8)()
for rnd in range(3):
    for name, L, h in libs:
        for _ in range(2): assert L.fqh_scan(h, buf.data_ptr(), n, 1, None, rs.data_ptr(), cap, summ, carry) == 0
        assert summ[0] == n // 330
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): L.fqh_scan(h, buf.data_ptr(), n, 1, None, rs.data_ptr(), cap, summ, carry)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10 * 1e3
        t = (C.c_float * 5)(); L.fqh_last_timing(h, t)
        print("%-24s exact path %.3f ms per step (index %.3f emit %.3f)" % (name, dt, t[1], t[3]), flush=True)
