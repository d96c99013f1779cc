"""tools/exp_alloc_order.py [N] — does the "kind" of an input allocation follow the ORDER in which the process
allocated it?  N inputs of 16 GiB, allocated one after the other and all kept alive; for each: the bare read (fqh_read_ceiling)
and the scan's index kernel (one context, adaptation off), min of 4.  Virtual addresses are printed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
n = (16 << 30) // 330 * 330
cap = n // 300 + 16
rs = torch.empty(cap, dtype=torch.int64, device=dev)
ctx = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.set_adapt_lines(0)
ctx.set_spin_wait(20000)
bufs = []
for i in range(N):
    b = torch.empty(n + 4096, dtype=torch.uint8, device=dev)
    ctx.synth_fill(b.data_ptr(), 0, n)
    bufs.append(b)
for rnd in range(2):
    for i, b in enumerate(bufs):
        rd = min(ctx.read_ceiling(b.data_ptr(), n)[1] for _ in range(3))
        best = 1e9
        for k in range(6):
            s = ctx.scan(b.data_ptr(), n, True, None, rs.data_ptr(), cap)[0]
            assert s.n_records == n // 330
            if k >= 2:
                best = min(best, ctx.timing().index_ms)
        print("round %d input %2d @ %#x: bare read %.3f ms, index kernel %.3f ms" % (rnd, i, b.data_ptr(), rd, best), flush=True)
