"""tools/exp_dirty.py LIB [LIB ...] — cold fqh_stats over 16 GiB of synthetic reads with one base in a million lower-cased and one
quality in a million '~', several builds of the library interleaved in ONE process: wall time per call, route, and the totals
(which must agree between the builds)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
GIB = float(os.environ.get("AB_GIB", "16"))
n = int(GIB * (1 << 30)) // 330 * 330
buf = torch.empty(n + 4096, dtype=torch.uint8, device=dev)
qh = torch.zeros(150 * 256, dtype=torch.int64, device=dev)
bh = torch.zeros(150 * 8, dtype=torch.int64, device=dev)
sc = torch.zeros(8, dtype=torch.int64, device=dev)
libs = []
for path in sys.argv[1:]:
    L = C.CDLL(os.path.abspath(path))
    h = C.c_void_p()
    L.fqh_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    assert L.fqh_create(0, C.byref(h)) == 0
    L.fqh_synth_fill.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
    L.fqh_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.fqh_invalidate.argtypes = [C.c_void_p]
    L.fqh_last_stats_route.argtypes = [C.c_void_p]
    libs.append((path, L, h))
assert libs[0][1].fqh_synth_fill(libs[0][2], buf.data_ptr(), 0, n, 0x5EEDF00D2026) == 0
gen = torch.Generator(device=dev); gen.manual_seed(1)
nrec = n // 330
nd = max(1, int(nrec * 150 * float(os.environ.get("AB_RATE", "1e-6"))))
ps = torch.unique(torch.randint(0, nrec, (nd,), device=dev, generator=gen) * 330 + 26 + torch.randint(0, 150, (nd,), device=dev, generator=gen))
pq = torch.unique(torch.randint(0, nrec, (nd,), device=dev, generator=gen) * 330 + 179 + torch.randint(0, 150, (nd,), device=dev, generator=gen))
buf[ps] = buf[ps] | 0x20
buf[pq] = 126
summ = (C.c_uint64 * 16)(); carry = (C.c_uint64 * 8)()
def step(L, h):
    L.fqh_invalidate(h)
    st = L.fqh_stats(h, buf.data_ptr(), n, 1, None, 150, qh.data_ptr(), bh.data_ptr(), sc.data_ptr(), summ, carry)
    assert st == 0, st
ref = None
K = 8
for rnd in range(3):
    for path, L, h in libs:
        for _ in range(2): step(L, h)
        qh.zero_(); bh.zero_(); sc.zero_()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(K): step(L, h)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K * 1e3
        tot = (int(qh.sum().item()), int(bh.sum().item()), int(bh.view(150, 8)[:, 5].sum().item()), int(qh.view(150, 256)[:, 126].sum().item()), sc.tolist())
        if ref is None: ref = tot
        print("%-24s %.3f ms per cold fqh_stats on dirty input (%d + %d bytes)  route %d  totals %s" % (os.path.basename(path), dt, ps.numel(), pq.numel(),
              L.fqh_last_stats_route(h), "ok" if tot == ref else "DIFFER %s vs %s" % (tot, ref)), flush=True)
assert ref[2] == K * ps.numel() and ref[3] == K * pq.numel(), ref
