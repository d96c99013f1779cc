"""tools/fuzz_sharded.py [seconds] [seed] — the sharded differential fuzz of tests/test_gpu_sharded_stream.py, for as long as one likes."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__ as g
from oracle import fqref
from test_gpu_sharded_stream import fuzz_sharded
pkg = g.load_package()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cases, errs = fuzz_sharded((torch, pkg, importlib.import_module("fastq_rs_amd.sharded")), fqref, seed, budget)
print("fuzz_sharded seed %d: %d files ok (%d with a parse error)" % (seed, cases, errs))
