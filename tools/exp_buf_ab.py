"""tools/exp_buf_ab.py LIB [LIB ...] — the 16 GiB scan step of several builds of the library on several INPUT allocations of one
process (the same bytes in each): which build is sensitive to where the input landed?"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
n = (16 << 30) // 330 * 330
cap = n // 300 + 16
NB = int(os.environ.get("AB_INPUTS", "5"))
rs = torch.empty(cap, dtype=torch.int64, device=dev)
bufs = [torch.empty(n + 4096, dtype=torch.uint8, device=dev) for _ in range(NB)]
libs = []
for path in sys.argv[1:]:
    L = C.CDLL(os.path.abspath(path))
    h = C.c_void_p()
    L.fqh_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    assert L.fqh_create(0, C.byref(h)) == 0
    L.fqh_synth_fill.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
    L.fqh_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    L.fqh_last_timing.argtypes = [C.c_void_p, C.c_void_p]
    libs.append((os.path.basename(path), L, h))
for b in bufs:
    assert libs[0][1].fqh_synth_fill(libs[0][2], b.data_ptr(), 0, n, 0x5EEDF00D2026) == 0
summ = (C.c_uint64 * 16)(); carry = (C.c_uint64 * 8)()
for rnd in range(2):
    for i, b in enumerate(bufs):
        row = []
        for name, L, h in libs:
            for _ in range(3): assert L.fqh_scan(h, b.data_ptr(), n, 1, None, rs.data_ptr(), cap, summ, carry) == 0
            assert summ[0] == n // 330
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(12): L.fqh_scan(h, b.data_ptr(), n, 1, None, rs.data_ptr(), cap, summ, carry)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 12 * 1e3
            t = (C.c_float * 5)(); L.fqh_last_timing(h, t)
            row.append("%s %.3f (index %.3f)" % (name, dt, t[1]))
        print("round %d input %d: %s" % (rnd, i, "   ".join(row)), flush=True)
