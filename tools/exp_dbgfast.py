import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import __graft_entry__ as g
import fuzzgen
pkg = g.load_package(); L = pkg.lib()
L.fqh_debug_fast_record.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
dev = torch.device("cuda:0")
rng = np.random.default_rng(1)
data = fuzzgen.valid_file(rng, 2000, seqlen=150)
n = len(data); nt = (n + 16383) // 16384
d = torch.empty(n + 16, dtype=torch.uint8, device=dev)
d[:n].copy_(torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()))
rs = torch.zeros(n // 6 + 3, dtype=torch.int64, device=dev)
recs = {}
for mode in ("old", "new"):
    if mode == "old": os.environ["FQH_FAST_OLD"] = "1"
    else: os.environ.pop("FQH_FAST_OLD", None)
    ctx = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    s, c, st = ctx.scan(d.data_ptr(), n, True, None, rs.data_ptr(), n // 6 + 3)
    print(mode, "fast:", ctx.last_scan_fast(), "records", s.n_records, "tiles", nt, "len", n)
    out = np.zeros((nt, 128), dtype=np.uint16)
    for t in range(nt):
        L.fqh_debug_fast_record(ctx._h, t, out[t].ctypes.data)
    recs[mode] = out
    ctx.close()
diff = np.argwhere(recs["old"] != recs["new"])
print("differing slots:", len(diff))
for t in sorted(set(diff[:, 0]))[:6]:
    cols = diff[diff[:, 0] == t][:, 1]
    print("tile", t, "cols", cols[:20], "old", recs["old"][t, cols[:10]], "new", recs["new"][t, cols[:10]], "meta old", recs["old"][t, 64:75], "meta new", recs["new"][t, 64:75])
