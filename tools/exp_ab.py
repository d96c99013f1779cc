"""tools/exp_ab.py — interleaved A/B of the fast-path index kernels inside one process."""
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
res = {"old": [], "new": []}
for rnd in range(8):
    for mode in ("old", "new"):
        if mode == "old": os.environ["FQH_FAST_OLD"] = "1"
        else: os.environ.pop("FQH_FAST_OLD", None)
        for _ in range(3):
            ctx.scan(buf.data_ptr(), nbytes, True, None, rs.data_ptr(), nrec + 16)
            assert ctx.last_scan_fast()
            t = ctx.timing()
            res[mode].append((t.index_ms, t.emit_ms, t.total_ms))
for mode in ("old", "new"):
    r = res[mode]
    print("%s: index min %.3f med %.3f | emit med %.3f | total min %.3f med %.3f" % (
        mode, min(x[0] for x in r), sorted(x[0] for x in r)[len(r) // 2], sorted(x[1] for x in r)[len(r) // 2],
        min(x[2] for x in r), sorted(x[2] for x in r)[len(r) // 2]), flush=True)
cs = sorted(ctx.read_ceiling(buf.data_ptr(), nbytes)[1] for _ in range(5))
print("ceiling min %.3f med %.3f" % (cs[0], cs[2]))
