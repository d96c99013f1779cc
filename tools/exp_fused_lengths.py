"""tools/exp_fused_lengths.py [GiB] — cold fqh_stats (single pass, k_scan_stats) by read length: is the single pass kept, and at what rate?
Realistic header lengths (Illumina-style, ~45 bytes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda:0")
GIB = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
rng = np.random.default_rng(7)
for L in (36, 50, 76, 100, 125, 150, 151, 200, 250):
    nrec = 4096
    seq = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), (nrec, L), p=[0.2495, 0.2495, 0.2495, 0.2495, 0.002])
    qual = rng.choice(np.frombuffer(b"#,5:F", dtype=np.uint8), (nrec, L))
    block = b"".join(b"@A00123:45:HXXXXXXXX:1:%04d:%05d:%05d 1:N:0:ATCACG\n" % (1101 + i % 400, 1000 + 7 * i, 2000 + 3 * i) + seq[i].tobytes() + b"\n+\n" +
                     qual[i].tobytes() + b"\n" for i in range(nrec))
    reps = int(GIB * (1 << 30)) // len(block)
    n = reps * len(block)
    buf = torch.from_numpy(np.frombuffer(block, dtype=np.uint8).copy()).to(dev).repeat(reps).contiguous()
    ctx = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
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
    assert int(sc[0]) == reps * nrec and int(qh.sum()) == reps * nrec * L and int(bh.sum()) == reps * nrec * L
    print("read length %3d (record %3d B): cold fqh_stats %.2f GiB in %.3f ms = %.0f GB/s, single pass kept: %s" % (
        L, len(block) // nrec, n / 2**30, best, n / 1e6 / best, ctx.last_scan_fast()), flush=True)
    ctx.close(); del buf
