"""tools/exp_shift.py VAR shift... — index time as a function of a workspace array's byte shift
(one process per shift: the shift is read once from the environment)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda:0")
nbytes = (16 << 30) // 330 * 330
nrec = nbytes // 330
ctx = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
buf = torch.empty(nbytes + 16, dtype=torch.uint8, device=dev)
ctx.synth_fill(buf.data_ptr(), 0, nbytes)
rs = torch.empty(nrec + 16, dtype=torch.int64, device=dev)
ts = []
for _ in range(10):
    ctx.scan(buf.data_ptr(), nbytes, True, None, rs.data_ptr(), nrec + 16)
    t = ctx.timing()
    ts.append((t.index_ms, t.emit_ms, t.total_ms))
print("%s index min %.3f med %.3f | emit min %.3f | total min %.3f" % (
    " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("FQH_WS")),
    min(x[0] for x in ts), sorted(x[0] for x in ts)[5], min(x[1] for x in ts), min(x[2] for x in ts)), flush=True)
