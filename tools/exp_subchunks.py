"""tools/exp_subchunks.py [K] — what sub-chunk pipelining inside one scan could buy: the 16 GiB buffer as K byte-range "shards"
on ONE GPU, each on a context and stream of its own, through the device-side protocol (prescan_launch -> words -> rescan_launch):
the emit / finalize kernels of shard k run while the byte scan of shard k+1 does.  Against the plain blocking fqh_scan."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda:0")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = (16 << 30) // 330 * 330
buf = torch.empty(n + 4096, dtype=torch.uint8, device=dev)
main = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
main.set_spin_wait(20000)
main.synth_fill(buf.data_ptr(), 0, n)
cap = n // 300 + 16
rs = torch.empty(cap, dtype=torch.int64, device=dev)
cut = [(n * k // K) // 16384 * 16384 for k in range(K)] + [n]
streams = [torch.cuda.Stream() for _ in range(K)]
ctxs = [pkg.Ctx(0, stream=s.cuda_stream) for s in streams]
for c in ctxs: c.set_spin_wait(20000)
W = pkg.SHARD_WORDS
words = torch.zeros(K * W, dtype=torch.int64, device=dev)
counts = torch.zeros(K * 2, dtype=torch.int64, device=dev)
rss = [torch.empty((cut[k + 1] - cut[k]) // 300 + 16, dtype=torch.int64, device=dev) for k in range(K)]
evs = [torch.cuda.Event() for _ in range(K)]
def serial(reps):
    for _ in range(reps):
        s, c, st = main.scan(buf.data_ptr(), n, True, None, rs.data_ptr(), cap)
        assert s.n_records == n // 330
def piped(reps):
    for _ in range(reps):
        for k in range(K):
            with torch.cuda.stream(streams[k]):
                ctxs[k].shard_prescan_launch(buf.data_ptr() + cut[k], cut[k + 1] - cut[k], words.data_ptr() + 8 * W * k)
                evs[k].record(streams[k])
        for k in range(K):
            with torch.cuda.stream(streams[k]):
                for j in range(k):
                    streams[k].wait_event(evs[j])
                ctxs[k].shard_rescan_launch(k == K - 1, words.data_ptr(), K, k, rss[k].data_ptr(), rss[k].numel(), counts.data_ptr() + 16 * k)
        tot = 0
        for k in range(K):
            s, c, st = ctxs[k].scan_finish()
            tot += s.n_records
        assert tot == n // 330, tot
for name, fn in (("serial fqh_scan", serial), ("%d sub-chunks, own streams" % K, piped), ("serial fqh_scan", serial), ("%d sub-chunks, own streams" % K, piped)):
    fn(4); torch.cuda.synchronize(); t0 = time.perf_counter(); R = 30
    fn(R); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / R * 1e3
    print("%-28s %.3f ms per 16 GiB  %.0f GB/s" % (name, dt, n / 1e6 / dt), flush=True)
