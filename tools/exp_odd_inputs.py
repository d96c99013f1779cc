"""tools/exp_odd_inputs.py — cold fqh_stats (4 GiB, 150 bp reads, lmax 150) on inputs the benchmarks do not hold: CRLF line ends,
'+id' separator lines, soft-masked (lower-case) bases in a share of the reads, qualities above '`' in a share of the reads."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda:0")
L, nrec = 150, 4096
def build(kind, share=0.0):
    rng = np.random.default_rng(7)
    seq = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), (nrec, L), p=[0.2495, 0.2495, 0.2495, 0.2495, 0.002])
    qual = rng.choice(np.frombuffer(b"#,5:F", dtype=np.uint8), (nrec, L))
    e = b"\r\n" if kind == "crlf" else b"\n"
    out = []
    for i in range(nrec):
        h = b"A00123:45:HXXXXXXXX:1:%04d:%05d:%05d 1:N:0:ATCACG" % (1101 + i % 400, 1000 + 7 * i, 2000 + 3 * i)
        s, q = seq[i].copy(), qual[i].copy()
        if kind == "lower" and rng.random() < share:
            a, b = sorted(rng.integers(0, L, 2)); s[a:b + 1] |= 0x20
        if kind == "highq" and rng.random() < share:
            q[rng.integers(0, L)] = 126
        out.append(b"@" + h + e + s.tobytes() + e + b"+" + (h if kind == "plusid" else b"") + e + q.tobytes() + e)
    return b"".join(out)
for kind, share in (("plain", 0), ("crlf", 0), ("plusid", 0), ("lower", 1e-3), ("lower", 0.05), ("lower", 1.0), ("highq", 1e-3), ("highq", 0.05), ("highq", 1.0)):
    block = build(kind, share)
    reps = (4 << 30) // len(block)
    n = reps * len(block)
    buf = torch.cat([torch.from_numpy(np.frombuffer(block, dtype=np.uint8).copy()).to(dev).repeat(reps), torch.zeros(16, dtype=torch.uint8, device=dev)])
    ctx = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    qh = torch.zeros(L * 256, dtype=torch.int64, device=dev); bh = torch.zeros(L * 8, dtype=torch.int64, device=dev); sc = torch.zeros(8, dtype=torch.int64, device=dev)
    ts = []
    for _ in range(5):
        qh.zero_(); bh.zero_(); sc.zero_(); ctx.invalidate(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.stats(buf.data_ptr(), n, L, qh.data_ptr(), bh.data_ptr(), sc.data_ptr())
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    assert int(sc[0]) == reps * nrec and int(qh.sum()) == reps * nrec * L
    print("%-7s share %-6g: calls %s ms (best %.0f GB/s); route %d" % (kind, share, " ".join("%.2f" % x for x in ts), n / 1e6 / min(ts), ctx.last_stats_route()), flush=True)
    ctx.close(); del buf
