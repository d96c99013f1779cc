"""tools/fuzz_tiny.py [seconds] [seed] — tiny files (0 .. 40 records of 0 .. 700 bases, any lmax from 1 to 1200, CRLF, truncation) through
fqh_stats on ONE context that keeps its history (rows hint, back-offs), against the oracle: the routing's edge cases rather than
the kernels'."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import __graft_entry__ as g
from oracle import fqref
pkg = g.load_package()
dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
t_end, cases, routes = time.time() + budget, 0, [0, 0, 0]
while time.time() < t_end:
    if cases % 50 == 49:
        ctx.close(); ctx = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    nrec = int(rng.integers(0, 41))
    L = int(rng.choice([0, 1, 3, 36, 64, 65, 150, 160, 161, 255, 256, 300, 511, 512, 513, 700]))
    rag = int(rng.choice([0, 0, 5, 400]))
    crlf = rng.random() < 0.2
    e = b"\r\n" if crlf else b"\n"
    recs = []
    for i in range(nrec):
        n = int(rng.integers(max(0, L - rag), L + 1))
        h = bytes(rng.integers(97, 123, int(rng.integers(1, 90))).astype(np.uint8).tolist())
        recs.append(b"@" + h + e + rng.choice(np.frombuffer(b"ACGTNacgt", dtype=np.uint8), n, p=[.24, .24, .24, .24, .02, .005, .005, .005, .005]).tobytes() + e + b"+" + e +
                    rng.integers(33, 127 if rng.random() < 0.1 else 75, n).astype(np.uint8).tobytes() + e)
    data = b"".join(recs)
    if data and rng.random() < 0.1:
        data = data[: len(data) - int(rng.integers(1, min(len(data), 50) + 1))]
    lmax = int(rng.choice([1, 2, 35, 36, 64, 100, 150, 151, 256, 300, 511, 512, 513, 700, 1200]))
    a = np.frombuffer(data, dtype=np.uint8)
    d = torch.zeros(a.size + 16, dtype=torch.uint8, device=dev)
    if a.size: d[: a.size].copy_(torch.from_numpy(a.copy()))
    qh = torch.zeros(lmax * 256, dtype=torch.int64, device=dev); bh = torch.zeros(lmax * 8, dtype=torch.int64, device=dev); sc = torch.zeros(8, dtype=torch.int64, device=dev)
    r, oq, ob, osc = fqref.stats(a, lmax)
    s, c = ctx.stats(d.data_ptr(), a.size, lmax, qh.data_ptr(), bh.data_ptr(), sc.data_ptr())
    assert (s.parse_status, s.n_records) == (r.status, r.n_records), ("status", cases, nrec, L, lmax, s.parse_status, s.n_records, r.status, r.n_records)
    assert np.array_equal(sc.cpu().numpy().astype(np.uint64), osc), ("scalars", cases, nrec, L, rag, lmax, sc.cpu().numpy(), osc)
    assert np.array_equal(qh.cpu().numpy().astype(np.uint64).reshape(lmax, 256), oq), ("qual", cases, nrec, L, rag, lmax)
    assert np.array_equal(bh.cpu().numpy().astype(np.uint64).reshape(lmax, 8), ob), ("base", cases, nrec, L, rag, lmax)
    routes[ctx.last_stats_route()] += 1
    cases += 1
print("fuzz_tiny: %d files ok (routes 0 / 1 / 2: %d / %d / %d)" % (cases, *routes))
