"""tools/exp_ab_env.py VAR A B — interleaved A/B of two values of an environment switch that the
library reads at every launch, inside one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
pkg = g.load_package()
var, va, vb = sys.argv[1], sys.argv[2], sys.argv[3]
dev = torch.device("cuda:0")
nbytes = (16 << 30) // 330 * 330
nrec = nbytes // 330
ctx = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
buf = torch.empty(nbytes + 16, dtype=torch.uint8, device=dev)
ctx.synth_fill(buf.data_ptr(), 0, nbytes)
rs = torch.empty(nrec + 16, dtype=torch.int64, device=dev)
res = {va: [], vb: []}
for rnd in range(8):
    for v in (va, vb):
        os.environ[var] = v
        for _ in range(3):
            s, c, st = ctx.scan(buf.data_ptr(), nbytes, True, None, rs.data_ptr(), nrec + 16)
            assert ctx.last_scan_fast() and s.n_records == nrec
            t = ctx.timing()
            res[v].append((t.index_ms, t.prefix_ms, t.emit_ms, t.total_ms))
for v in (va, vb):
    r = res[v]
    med = lambda k: sorted(x[k] for x in r)[len(r) // 2]
    print("%s=%s: index min %.3f med %.3f | prefix med %.3f | emit med %.3f | total min %.3f med %.3f" % (
        var, v, min(x[0] for x in r), med(0), med(1), med(2), min(x[3] for x in r), med(3)), flush=True)
