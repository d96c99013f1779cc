#!/usr/bin/env python3
"""Knock-out timing of k_scan_stats (FQH_FZ_DBG flags; results are wrong by design; needs a library built with -DFQH_TUNING:
make -C fastq-rs_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC -DFQH_TUNING").  usage: exp_fzdbg.py flags..."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, time, torch
sys.path.insert(0, %r)
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda:0")
ctx = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
n = (16 << 30) // 330 * 330
buf = torch.empty(n + 16, dtype=torch.uint8, device=dev)
ctx.synth_fill(buf.data_ptr(), 0, n)
qh = torch.zeros(150 * 256, dtype=torch.int64, device=dev); bh = torch.zeros(150 * 8, dtype=torch.int64, device=dev); sc = torch.zeros(8, dtype=torch.int64, device=dev)
ts = []
for i in range(5):
    ctx.set_spec(True)
    ctx.stats(buf.data_ptr(), n, 150, qh.data_ptr(), bh.data_ptr(), sc.data_ptr())
    ts.append(ctx.timing().index_ms)
print("FQH_FZ_DBG=%%s kernel ms: %%s fast %%s" %% (os.environ.get("FQH_FZ_DBG"), " ".join("%%.3f" %% t for t in ts), ctx.last_scan_fast()), flush=True)
''' % ROOT
for f in sys.argv[1:]:
    env = dict(os.environ, FQH_FZ_DBG=f)
    subprocess.run([sys.executable, "-c", code], env=env)
