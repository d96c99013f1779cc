"""tools/ab_stats.py LIB [LIB ...] — cold fqh_stats (single pass: k_scan_stats) over 16 GiB with several builds of the library in ONE
process and on ONE box, interleaved; the totals of every build must agree with the first one's."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
GIB = float(os.environ.get("AB_GIB", "16"))
LMAX = int(os.environ.get("AB_LMAX", "150"))
n = int(GIB * (1 << 30)) // 330 * 330
buf = torch.empty(n + 4096, dtype=torch.uint8, device=dev)
qh = torch.zeros(LMAX * 256, dtype=torch.int64, device=dev)
bh = torch.zeros(LMAX * 8, dtype=torch.int64, device=dev)
sc = torch.zeros(8, dtype=torch.int64, device=dev)
libs = []
for path in sys.argv[1:]:
    L = C.CDLL(os.path.abspath(path))
    h = C.c_void_p()
    L.fqh_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    assert L.fqh_create(0, C.byref(h)) == 0
    L.fqh_synth_fill.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
    L.fqh_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                            C.c_void_p, C.c_void_p]
    L.fqh_invalidate.argtypes = [C.c_void_p]
    L.fqh_last_scan_fast.argtypes = [C.c_void_p]
    libs.append((path, L, h))
assert libs[0][1].fqh_synth_fill(libs[0][2], buf.data_ptr(), 0, n, 0x5EEDF00D2026) == 0
summ = (C.c_uint64 * 16)(); carry = (C.c_uint64 * 8)()
def step(L, h):
    L.fqh_invalidate(h)
    st = L.fqh_stats(h, buf.data_ptr(), n, 1, None, LMAX, qh.data_ptr(), bh.data_ptr(), sc.data_ptr(), summ, carry)
    assert st == 0, st
ref = None
for rnd in range(3):
    for path, L, h in libs:
        for _ in range(2): step(L, h)
        qh.zero_(); bh.zero_(); sc.zero_()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        K = 10
        for _ in range(K): step(L, h)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K * 1e3
        tot = (int(qh.sum().item()), int(bh.sum().item()), int((qh * torch.arange(qh.numel(), device=dev)).sum().item()) & 0xFFFFFFFFFFFF, sc.tolist())
        if ref is None: ref = tot
        ok = tot == ref
        print("%-32s %.3f ms per cold fqh_stats  %.0f GB/s  fast=%d  totals %s" % (os.path.basename(path), dt, n / 1e6 / dt,
              L.fqh_last_scan_fast(h), "ok" if ok else "DIFFER %s vs %s" % (tot, ref)), flush=True)
