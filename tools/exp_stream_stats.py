"""tools/exp_stream_stats.py [LMAX ..] — the pinned ring with histograms per slot (FQH_STREAM_STATS) over slots that are filled once and
submitted again and again (no producer: what the DEVICE side of a streamed statistics run costs per slot), 150-base reads."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda:0")
rng = np.random.default_rng(5)
L, nrec = 150, 4096
seq = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), (nrec, L), p=[0.2495, 0.2495, 0.2495, 0.2495, 0.002])
qual = rng.choice(np.frombuffer(b"#,5:F", dtype=np.uint8), (nrec, L))
block = b"".join(b"@A00123:45:HXXXXXXXX:1:%04d:%05d:%05d 1:N:0:ATCACG\n" % (1101 + i % 400, 1000 + 7 * i, 2000 + 3 * i) + seq[i].tobytes() + b"\n+\n" + qual[i].tobytes() + b"\n" for i in range(nrec))
slot = int(os.environ.get("SLOT_MIB", "255")) << 20
reps = slot // len(block)
fill = np.frombuffer(block * reps, dtype=np.uint8)
for lmax in [int(x) for x in sys.argv[1:]] or [150, 1000]:
    ctx = pkg.Ctx(0)
    st = pkg.Stream(ctx, slot, 3, pkg.STREAM_STATS | pkg.STREAM_TIMING)
    qh = torch.zeros(lmax * 256, dtype=torch.int64, device=dev); bh = torch.zeros(lmax * 8, dtype=torch.int64, device=dev); sc = torch.zeros(8, dtype=torch.int64, device=dev)
    st.set_stats(lmax, qh.data_ptr(), bh.data_ptr(), sc.data_ptr())
    n_slots, filled, sub, col = 48, set(), 0, 0
    t0 = time.perf_counter()
    while col < n_slots:
        a = st.acquire() if sub < n_slots else None
        if a is not None:
            p, cap = a
            if p not in filled:
                C.memmove(p, fill.ctypes.data, fill.size)
                filled.add(p)
                if len(filled) == 3: t0 = time.perf_counter()   # (the clock starts when the three slots are filled)
            st.submit(fill.size, sub == n_slots - 1)
            sub += 1
        else:
            c = st.collect(); st.release(); col += 1
    torch.cuda.synchronize()
    w = time.perf_counter() - t0
    t = st.timing()
    assert int(sc[0]) == n_slots * reps * nrec and int(qh.sum()) == n_slots * reps * nrec * L
    print("lmax %4d: %d slots of %.0f MiB, wall %.1f ms after the fill (%.1f GB/s), copy busy %.1f ms, scan busy %.1f ms (%.3f ms per slot)" % (
        lmax, n_slots, fill.size / 2**20, w * 1e3, (n_slots - 3) * fill.size / 1e9 / w, t.copy_busy_ms, t.scan_busy_ms, t.scan_busy_ms / n_slots), flush=True)
    st.close(); ctx.close()
