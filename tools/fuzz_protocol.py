"""tools/fuzz_protocol.py [seconds] [seed] — the differential fuzz of tests/test_gpu_fuzz_protocol.py, for as long as one likes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import __graft_entry__ as g
from oracle import fqref
from test_gpu_fuzz_protocol import fuzz_protocol
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cases, agains, errs = fuzz_protocol(torch, g.load_package(), fqref, seed, budget)
print("fuzz_protocol seed %d: %d files ok (%d through the host recipe, %d with a parse error)" % (seed, cases, agains, errs))
