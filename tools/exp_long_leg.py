"""tools/exp_long_leg.py LIB — bench.py's two long-read legs under another build of the library (FQH_LIB_PATH)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["FQH_LIB_PATH"] = os.path.abspath(sys.argv[1])
import torch
import __graft_entry__ as g
import bench
pkg = g.load_package()
dev = torch.device("cuda:0")
ctx = pkg.Ctx(0)
for varied in (False, True):
    r = bench.long_read_leg(pkg, torch, dev, ctx, varied=varied)
    print(os.path.basename(sys.argv[1]), "varied" if varied else "fixed", r["end_to_end_ms"], r["histogram_kernels_ms"], r["gbs_end_to_end"], r["gbs_histograms"],
          os.environ.get("FQH_LONG_ROUNDS", ""))
