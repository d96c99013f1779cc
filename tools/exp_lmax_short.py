"""tools/exp_lmax_short.py READ_LEN LMAX [LMAX ..] — cold fqh_stats of 4 GiB of reads of READ_LEN bases with the caller's rows cut at LMAX
(the columns beyond go to the overflow counters): end-to-end time, route, kernel times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda:0")
L = int(sys.argv[1])
rng = np.random.default_rng(7)
nrec = 1024
seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), (nrec, L))
qual = rng.choice(np.frombuffer(b"#,5:F", dtype=np.uint8), (nrec, L))
block = b"".join(b"@r%06d\n" % i + seq[i].tobytes() + b"\n+\n" + qual[i].tobytes() + b"\n" for i in range(nrec))
reps = (4 << 30) // len(block)
n = reps * len(block)
d = torch.cat([torch.from_numpy(np.frombuffer(block, dtype=np.uint8).copy()).to(dev).repeat(reps), torch.zeros(16, dtype=torch.uint8, device=dev)])
for lmax in [int(x) for x in sys.argv[2:]]:
    ctx = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    qh = torch.zeros(lmax * 256, dtype=torch.int64, device=dev); bh = torch.zeros(lmax * 8, dtype=torch.int64, device=dev); sc = torch.zeros(8, dtype=torch.int64, device=dev)
    ts = []
    for _ in range(5):
        qh.zero_(); bh.zero_(); sc.zero_(); ctx.invalidate(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.stats(d.data_ptr(), n, lmax, qh.data_ptr(), bh.data_ptr(), sc.data_ptr())
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    assert int(sc[0]) == reps * nrec and int(qh.sum()) == reps * nrec * min(L, lmax)
    t = ctx.timing()
    print("reads of %d bp, lmax %d: calls %s ms (best %.0f GB/s); route %d, kernels: index %.3f stats %.3f" % (
        L, lmax, " ".join("%.2f" % x for x in ts), n / 1e6 / min(ts), ctx.last_stats_route(), t.index_ms, t.stats_ms), flush=True)
    ctx.close()
