"""tools/ab_scan.py LIB_A LIB_B — the 16 GiB scan step (fqh_scan with offsets) with two builds of the library in ONE process and
on ONE box, interleaved: what a change to the scan kernels costs, free of box-to-box variance."""
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
    libs.append((path, L, h))
assert libs[0][1].fqh_synth_fill(libs[0][2], buf.data_ptr(), 0, n, 0x5EEDF00D2026) == 0
summ = (C.c_uint64 * 16)(); carry = (C.c_uint64 * 8)()
def step(L, h):
    assert L.fqh_scan(h, buf.data_ptr(), n, 1, None, rs.data_ptr(), cap, summ, carry) in (0,)
    assert summ[0] == n // 330
for rnd in range(4):
    for path, L, h in libs:
        for _ in range(3): step(L, h)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): step(L, h)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20 * 1e3
        t = (C.c_float * 5)()
        L.fqh_last_timing.argtypes = [C.c_void_p, C.c_void_p]
        L.fqh_last_timing(h, t)
        print("%-40s %.3f ms per step  %.0f GB/s   (last step: index %.3f prefix %.3f emit+finalize %.3f)" % (
            os.path.basename(path), dt, n / 1e6 / dt, t[1], t[2], t[3]), flush=True)
