"""tools/exp_filter.py — scan -> index -> flags -> gather on 8 GiB of the synthetic file (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda:0")
nbytes = (8 << 30) // 330 * 330
n = nbytes // 330
ctx = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
buf = torch.empty(nbytes + 16, dtype=torch.uint8, device=dev)
ctx.synth_fill(buf.data_ptr(), 0, nbytes)
rs = torch.empty(n + 1, dtype=torch.int64, device=dev)
idx = torch.empty(n * 24, dtype=torch.uint8, device=dev)
flags = torch.empty(n, dtype=torch.uint8, device=dev)
out = torch.empty(nbytes, dtype=torch.uint8, device=dev)
def t(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3
ms_scan = t(lambda: ctx.scan(buf.data_ptr(), nbytes, True, None, rs.data_ptr(), n + 1))
ms_idx = t(lambda: ctx.index_records(idx.data_ptr(), n))
ms_flags = t(lambda: ctx.record_flags(buf.data_ptr(), nbytes, idx.data_ptr(), n, flags.data_ptr()))
res = {}
def gather():
    res["r"] = ctx.gather_records(buf.data_ptr(), nbytes, idx.data_ptr(), n, flags.data_ptr(), 3, 3, out.data_ptr(), nbytes)
ms_gather = t(gather)
st, ns, nb = res["r"]
print("8 GiB, %d records: scan %.2f ms, index_records %.2f ms, flags %.2f ms, gather %.2f ms (%d records, %.2f GiB kept: reads without N)" % (
    n, ms_scan, ms_idx, ms_flags, ms_gather, ns, nb / 2**30))
print("pipeline %.1f GB/s of input" % (nbytes / 1e6 / (ms_scan + ms_idx + ms_flags + ms_gather)))
