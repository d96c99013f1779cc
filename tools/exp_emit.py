"""tools/exp_emit.py — emit-stage timing with and without the record-offset stores (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda:0")
ctx = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
nbytes = (16 << 30) // 330 * 330
buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
ctx.synth_fill(buf.data_ptr(), 0, nbytes)
nrec = nbytes // 330
rs = torch.empty(nrec + 1, dtype=torch.int64, device=dev)
for mode in ("offsets", "count-only", "offsets", "count-only"):
    ts = []
    for _ in range(8):
        if mode == "offsets":
            ctx.scan(buf.data_ptr(), nbytes, True, None, rs.data_ptr(), nrec + 1)
        else:
            ctx.scan(buf.data_ptr(), nbytes, True, None, None, 0)
        t = ctx.timing()
        ts.append((t.emit_ms, t.index_ms, t.total_ms))
    ts.sort()
    print(mode, "emit min %.3f med %.3f | index min %.3f | total min %.3f" % (ts[0][0], ts[4][0], min(x[1] for x in ts), min(x[2] for x in ts)), flush=True)
