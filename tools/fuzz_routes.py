"""tools/fuzz_routes.py [seconds] [seed] — the differential route fuzz of tests/test_gpu_fuzz_routes.py, for as long as one likes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__ as g
from oracle import fqref
from test_gpu_fuzz_routes import fuzz_routes
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cases, fast, fused = fuzz_routes(torch, g.load_package(), fqref, seed, budget)
print("fuzz_routes seed %d: %d files ok (%d scans kept the fast path, %d statistics calls the single pass)" % (seed, cases, fast, fused))
