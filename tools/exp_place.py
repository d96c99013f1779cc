"""tools/exp_place.py — does the index kernel's time depend on where the buffers sit in HBM?
Same process, same box: allocate a pad of varying size first, then the input, the offsets and a fresh
context (workspace), scan, report."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda:0")
nbytes = (16 << 30) // 330 * 330
nrec = nbytes // 330
for pad_mb in (0, 0, 1, 64, 1000, 4096, 16384, 40000, 0, 3):
    pad = torch.empty(pad_mb << 20, dtype=torch.uint8, device=dev) if pad_mb else None
    ctx = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    buf = torch.empty(nbytes + 16, dtype=torch.uint8, device=dev)
    ctx.synth_fill(buf.data_ptr(), 0, nbytes)
    rs = torch.empty(nrec + 16, dtype=torch.int64, device=dev)
    ts = []
    for _ in range(8):
        ctx.scan(buf.data_ptr(), nbytes, True, None, rs.data_ptr(), nrec + 16)
        t = ctx.timing()
        ts.append((t.index_ms, t.emit_ms, t.total_ms))
    cs = min(ctx.read_ceiling(buf.data_ptr(), nbytes)[1] for _ in range(3))
    print("pad %6d MB buf@%x rs@%x: index min %.3f med %.3f | emit min %.3f | total min %.3f | ceiling %.3f" % (
        pad_mb, buf.data_ptr(), rs.data_ptr(), min(x[0] for x in ts), sorted(x[0] for x in ts)[4],
        min(x[1] for x in ts), min(x[2] for x in ts), cs), flush=True)
    ctx.close()
    del buf, rs, pad, ctx
    torch.cuda.empty_cache()
