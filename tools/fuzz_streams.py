"""tools/fuzz_streams.py [seconds] [seed] — the differential fuzz of tests/test_gpu_fuzz_streams.py, for as long as one likes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__ as g
from oracle import fqref
from test_gpu_fuzz_streams import fuzz_streams
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cases, single = fuzz_streams(torch, g.load_package(), fqref, seed, budget)
print("fuzz_streams seed %d: %d files ok (%d slots / chunks kept the single pass)" % (seed, cases, single))
