#!/usr/bin/env python3
"""One warm-up + N timed single-pass launches at a given size (for rocprofv3 PMC passes with FQH_FZ_DBG knock-outs).
usage: exp_fzone.py [GiB] [launches]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda:0")
ctx = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n = int(gib * (1 << 30)) // 330 * 330
buf = torch.empty(n + 16, dtype=torch.uint8, device=dev)
ctx.synth_fill(buf.data_ptr(), 0, n)
qh = torch.zeros(150 * 256, dtype=torch.int64, device=dev); bh = torch.zeros(150 * 8, dtype=torch.int64, device=dev); sc = torch.zeros(8, dtype=torch.int64, device=dev)
ts = []
for i in range(reps):
    ctx.set_spec(True)
    ctx.stats(buf.data_ptr(), n, 150, qh.data_ptr(), bh.data_ptr(), sc.data_ptr())
    ts.append(ctx.timing().index_ms)
print("FQH_FZ_DBG=%s %.2f GiB kernel ms: %s fast %s" % (os.environ.get("FQH_FZ_DBG"), gib, " ".join("%.3f" % t for t in ts), ctx.last_scan_fast()), flush=True)
