#!/usr/bin/env python3
"""Single-pass scan + histograms (k_scan_stats): parity against the oracle on a sample, then timing at full size.
usage: exp_fused.py [GiB]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
from oracle import fqref
pkg = g.load_package()
dev = torch.device("cuda:0")
ctx = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
LMAX = 150

def run(buf, n, want_offsets=False):
    qh = torch.zeros(LMAX * 256, dtype=torch.int64, device=dev)
    bh = torch.zeros(LMAX * 8, dtype=torch.int64, device=dev)
    sc = torch.zeros(8, dtype=torch.int64, device=dev)
    rs = torch.zeros(n // 300 + 16, dtype=torch.int64, device=dev) if want_offsets else None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if want_offsets:
        s, c, st = ctx.scan_stats(buf.data_ptr(), n, LMAX, qh.data_ptr(), bh.data_ptr(), sc.data_ptr(),
                                  d_rec_start=rs.data_ptr(), cap=rs.numel())
    else:
        s, c = ctx.stats(buf.data_ptr(), n, LMAX, qh.data_ptr(), bh.data_ptr(), sc.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    return s, qh, bh, sc, rs, dt

# parity on 64 MiB + a partial tile
n = (64 << 20) // 330 * 330 + 0
buf = torch.empty(n + 16, dtype=torch.uint8, device=dev)
ctx.synth_fill(buf.data_ptr(), 0, n)
host = buf[:n].cpu().numpy()
for wo in (False, True):
    s, qh, bh, sc, rs, dt = run(buf, n, wo)
    r2, oq, ob, osc = fqref.stats(host, LMAX)
    print("offsets" if wo else "stats  ", "fast path kept:", ctx.last_scan_fast(), "status", s.parse_status, "records", s.n_records, r2.n_records,
          "qual ok", np.array_equal(qh.cpu().numpy().astype(np.uint64).reshape(LMAX, 256), oq),
          "base ok", np.array_equal(bh.cpu().numpy().astype(np.uint64).reshape(LMAX, 8), ob),
          "scalars", sc.cpu().numpy().tolist(), osc.tolist(), "%.3f ms" % dt, flush=True)
    if wo:
        r, off = fqref.offsets(host)
        print("   offsets ok", np.array_equal(rs.cpu().numpy()[: r.n_records].astype(np.uint64), off))
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 16
n = int(gib * (1 << 30)) // 330 * 330
del buf
buf = torch.empty(n + 16, dtype=torch.uint8, device=dev)
ctx.synth_fill(buf.data_ptr(), 0, n)
for wo in (False, True):
    for i in range(4):
        s, qh, bh, sc, rs, dt = run(buf, n, wo)
        t = ctx.timing()
        print("%s %.2f GiB: wall %.3f ms | kernel(index) %.3f prefix %.3f emit %.3f total %.3f | fast %s records %d sum(q) %d sum(b) %d sc %s" % (
            "offsets+stats" if wo else "stats", n / 2**30, dt, t.index_ms, t.prefix_ms, t.emit_ms, t.total_ms, ctx.last_scan_fast(), s.n_records,
            int(qh.sum().item()), int(bh.sum().item()), sc.cpu().numpy().tolist()[:5]), flush=True)
    assert int(sc[0].item()) == n // 330 and int(qh.sum().item()) == n // 330 * 150
