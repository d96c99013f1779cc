"""tools/ab_fused.py [GiB] — cold fqh_stats (single pass) on synthetic 150 bp input, A/B of FQH_FUSED_V in separate processes
is done by the caller (env); prints best-of-N kernel and wall times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
pkg = g.load_package()
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
n = int(gib * (1 << 30)) // 330 * 330
d = torch.empty(n + 16, dtype=torch.uint8, device=dev)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
ctx = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.set_spin_wait(20000)
ctx.synth_fill(d.data_ptr(), 0, n)
qh = torch.zeros(150 * 256, dtype=torch.int64, device=dev); bh = torch.zeros(150 * 8, dtype=torch.int64, device=dev)
sc = torch.zeros(8, dtype=torch.int64, device=dev)
best = None
for i in range(8):
    qh.zero_(); bh.zero_(); sc.zero_(); ctx.invalidate(); torch.cuda.synchronize()
    t = time.perf_counter()
    ctx.stats(d.data_ptr(), n, 150, qh.data_ptr(), bh.data_ptr(), sc.data_ptr())
    torch.cuda.synchronize()
    w = (time.perf_counter() - t) * 1e3
    tt = ctx.timing()
    assert int(sc[0].item()) == n // 330 and int(qh.sum().item()) == n // 330 * 150 == int(bh.sum().item()), (sc, qh.sum(), bh.sum())
    if best is None or w < best[0]: best = (w, tt.index_ms, ctx.last_scan_fast())
print("FQH_FUSED_V=%s: wall %.3f ms, kernel %.3f ms, single pass kept: %s" % (os.environ.get("FQH_FUSED_V", "2"), best[0], best[1], best[2]))
