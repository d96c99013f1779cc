"""tools/sweep_read_length.py [GiB] [lengths ..] (RAGGED=lo: every record's length uniform in lo .. length, trimmed reads) — cold fqh_stats by read length: Illumina-style ids, qualities of
five instrument bins, N at 0.2 %; the route each length takes (fqh_last_stats_route: 1 / 2 the scan's own pass, 0 a second pass),
end-to-end time of the blocking call (best of 5) and GB/s.  The reference treats every record up to BUFSIZE alike
(src/records.rs:75-90, src/lib.rs:276-283): no length should fall off a cliff."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda:0")
GIB = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
LENS = [int(x) for x in sys.argv[2:]] or [36, 50, 76, 100, 151, 250, 300, 500]
rng = np.random.default_rng(7)
for L in LENS:
    nrec = 4096
    seq = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), (nrec, L), p=[0.2495, 0.2495, 0.2495, 0.2495, 0.002])
    qual = rng.choice(np.frombuffer(b"#,5:F", dtype=np.uint8), (nrec, L))
    lo = int(os.environ.get("RAGGED", "0"))
    lens = rng.integers(min(lo, L), L + 1, nrec) if lo else np.full(nrec, L)
    if os.environ.get("RAGGED_SET"):   # lengths drawn from a set (<= L), e.g. RAGGED_SET=144,148
        lens = rng.choice(np.array([int(x) for x in os.environ["RAGGED_SET"].split(",")]), nrec)
        lo = int(lens.min())
    if lo:   # (columns beyond a record's length do not count)
        seq[np.arange(L)[None, :] >= lens[:, None]] = 0
    block = b"".join(b"@A00123:45:HXXXXXXXX:1:%04d:%05d:%05d 1:N:0:ATCACG\n" % (1101 + i % 400, 1000 + 7 * i, 2000 + 3 * i) + seq[i, :lens[i]].tobytes() + b"\n+\n" +
                     qual[i, :lens[i]].tobytes() + b"\n" for i in range(nrec))
    reps = int(GIB * (1 << 30)) // len(block)
    n = reps * len(block)
    buf = torch.cat([torch.from_numpy(np.frombuffer(block, dtype=np.uint8).copy()).to(dev).repeat(reps), torch.zeros(16, dtype=torch.uint8, device=dev)])
    ctx = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    ctx.set_spin_wait(20000)
    qh = torch.zeros(L * 256, dtype=torch.int64, device=dev); bh = torch.zeros(L * 8, dtype=torch.int64, device=dev)
    sc = torch.zeros(8, dtype=torch.int64, device=dev)
    best = None
    for _ in range(5):
        qh.zero_(); bh.zero_(); sc.zero_(); ctx.invalidate(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.stats(buf.data_ptr(), n, L, qh.data_ptr(), bh.data_ptr(), sc.data_ptr())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        best = dt if best is None else min(best, dt)
    assert int(sc[0]) == reps * nrec and int(qh.sum()) == reps * int(lens.sum()) and int(bh.sum()) == reps * int(lens.sum())
    exp_b = np.stack([(seq == c).sum(axis=0) for c in b"ACGTN"], axis=1) * reps
    assert np.array_equal(bh.cpu().numpy().reshape(L, 8)[:, :5], exp_b), "base histogram differs from the host count"
    route = ctx.last_stats_route()
    t = ctx.timing()
    print("read length %s%3d (record %3d B): cold fqh_stats %.2f GiB in %.3f ms = %4.0f GB/s  route %d (%s), scan on the fast path: %s; kernels: index %.3f stats %.3f total %.3f ms" % (
        ("%d .. " % lo) if lo else "", L, len(block) // nrec, n / 2**30, best, n / 1e6 / best, route,
        {1: "single pass", 2: "single pass + declined lines behind it", 0: "second pass"}[route], bool(ctx.last_scan_fast()),
        t.index_ms, t.stats_ms, t.total_ms), flush=True)
    ctx.close(); del buf
