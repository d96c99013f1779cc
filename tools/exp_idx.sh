#!/bin/bash
# tools/exp_idx.sh "<defs>" .. — tuning builds of the library with extra -D switches (knock-outs of k_index_fast: wrong results by design,
# the fall-back is suppressed with FQH_FZ_DBG=1), index kernel time of bench.py's step for each
cd "$(dirname "$0")/.."
mkdir -p /tmp/tuning
for defs in "$@"; do
(cd fastq-rs_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DFQH_TUNING $defs -shared -o /tmp/tuning/libfastq_hip_i.so *.hip -ldl 2>&1 | grep -E "error")
echo "== $defs"
FQH_FZ_DBG=1 FQH_LIB_PATH=/tmp/tuning/libfastq_hip_i.so python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda:0")
n = (16 << 30) // 330 * 330
buf = torch.empty(n + 4096, dtype=torch.uint8, device=dev)
ctx = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.set_spin_wait(20000)
ctx.synth_fill(buf.data_ptr(), 0, n)
rs = torch.empty(n // 300 + 16, dtype=torch.int64, device=dev)
ts = []
for i in range(12):
    s, c, st = ctx.scan(buf.data_ptr(), n, True, None, rs.data_ptr(), rs.numel())
    t = ctx.timing(); ts.append((t.index_ms, t.total_ms))
ts = ts[2:]
print("index %.3f (min %.3f) total %.3f ms  records %d fast %s" % (sum(x[0] for x in ts) / len(ts), min(x[0] for x in ts), sum(x[1] for x in ts) / len(ts), s.n_records, ctx.last_scan_fast()))
PY
done
