"""tools/exp_flags.py — index-kernel time with individual output stores disabled (timing only: the
scan then falls back or reports garbage; only timing().index_ms of the speculative launch is read)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
pkg = g.load_package()
L = pkg.lib()
dev = torch.device("cuda:0")
nbytes = (16 << 30) // 330 * 330
nrec = nbytes // 330
ctx = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
buf = torch.empty(nbytes + 16, dtype=torch.uint8, device=dev)
ctx.synth_fill(buf.data_ptr(), 0, nbytes)
rs = torch.empty(nrec + 16, dtype=torch.int64, device=dev)
ctx.scan(buf.data_ptr(), nbytes, True, None, rs.data_ptr(), nrec + 16)
for rnd in range(3):
    for f in [int(x) for x in os.environ.get("FQH_EXP_FLAGS", "0,4,8,16,32,0").split(",")]:
        L.fqh_debug_set_flags(ctypes.c_uint(f))
        ts = []; te = []
        for _ in range(5):  # the skipped stores' targets still hold the previous (identical) results
            ctx.scan(buf.data_ptr(), nbytes, True, None, rs.data_ptr(), nrec + 16)
            assert ctx.last_scan_fast()
            tt = ctx.timing(); ts.append(tt.index_ms); te.append(tt.emit_ms)
        print("flags %d: index min %.3f med %.3f | emit min %.3f med %.3f" % (f, min(ts), sorted(ts)[2], min(te), sorted(te)[2]), flush=True)
L.fqh_debug_set_flags(ctypes.c_uint(0))
