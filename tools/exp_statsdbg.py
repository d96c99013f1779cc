"""tools/exp_statsdbg.py [GiB] — histogram kernel time on the bench's synthetic 150 bp data; with
FQH_STATS_DBG set (bits: see StatsArgs::dbg in csrc/fqh_internal.h) the results are wrong by design
and are not checked: this is the time decomposition of DESIGN.md §5."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda:0")
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 16.0
RECLEN = 330
n = int(gib * 2**30) // RECLEN * RECLEN
buf = torch.empty(n + 4096, dtype=torch.uint8, device=dev)
ctx = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.synth_fill(buf.data_ptr(), 0, n)
qh = torch.zeros(150 * 256, dtype=torch.int64, device=dev); bh = torch.zeros(150 * 8, dtype=torch.int64, device=dev)
sc = torch.zeros(8, dtype=torch.int64, device=dev)
ts = []
for _ in range(6):
    qh.zero_(); bh.zero_(); sc.zero_()
    ctx.stats_launch(buf.data_ptr(), n, 150, qh.data_ptr(), bh.data_ptr(), sc.data_ptr()); ctx.stats_finish()
    ts.append(ctx.timing().stats_ms)
print("FQH_STATS_DBG=%s: %.2f GiB  stats kernel %.3f ms (min of 6; all: %s)  %.0f GB/s  qual entries %d" % (
    os.environ.get("FQH_STATS_DBG", "0"), n / 2**30, min(ts), " ".join("%.3f" % t for t in ts), n / 1e6 / min(ts), int(qh.sum())))
if int(os.environ.get("FQH_STATS_DBG", "0")) & 8192:  # cycles per wave and launch in the kernel's sections (4096 waves)
    print("  cycles per wave: words %.0f  staging %.0f  lines %.0f  batches %.0f" % tuple(int(x) / 4096.0 for x in qh[:4]))
