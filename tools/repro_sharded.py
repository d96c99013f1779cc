"""tools/repro_sharded.py SEED [seconds] — fuzz_sharded of tests/test_gpu_sharded_stream.py, but a failing case is written to
gpurun_out/repro_sharded_SEED.npz (data, cuts, slot size) with what every shard reported, instead of ending the run."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import __graft_entry__ as g
import fuzzgen
from test_gpu_sharded_stream import run_sharded
from oracle import fqref
pkg = g.load_package()
env = (torch, pkg, importlib.import_module("fastq_rs_amd.sharded"))
seed = int(sys.argv[1]); budget = float(sys.argv[2]) if len(sys.argv) > 2 else 240.0
rng = np.random.default_rng(seed)
t_end = time.time() + budget
cases = 0
while time.time() < t_end:
    L = int(rng.choice([20, 75, 150, 300]))
    data = fuzzgen.valid_file(rng, int(rng.integers(300, 12000)), maxlen=L, crlf=bool(rng.random() < 0.15))
    kind = rng.random()
    if kind < 0.25:
        data = fuzzgen.mutate(rng, data, 1)
    elif kind < 0.35:
        data = data[: len(data) - int(rng.integers(1, 300))]
    n = len(data)
    k = int(rng.integers(1, 5))
    cuts = sorted(set(int(x) for x in rng.integers(1, n, k)))
    if rng.random() < 0.3:    # (as fuzz_sharded: two cuts close together, a cut right behind a newline)
        c0 = cuts[int(rng.integers(0, len(cuts)))]
        cuts = sorted(set(cuts + [min(n - 1, c0 + int(rng.integers(1, 400)))]))
    if rng.random() < 0.2:
        j = data.find(b"\n", cuts[0])
        if 0 < j + 1 < n:
            cuts = sorted(set(cuts + [j + 1]))
    slot = int(rng.choice([1 << 16, 1 << 18, 1 << 20]))
    try:
        status, n_records, hist, shards = run_sharded(env, data, cuts, 150, slot_bytes=slot)
    except pkg.FqhError as e:
        continue
    r, oq, ob, osc = fqref.stats(data, 150)
    bad = (r.status == pkg.OK and (status, n_records) != (pkg.OK, r.n_records))
    window_error = any(sh.res.status == pkg.E_HEADER and sh.res.n_records == 0 and sh.lo > 0 for sh in shards)
    bad = bad or (r.status != pkg.OK and not window_error and (status, n_records) != (r.status, r.n_records))
    if bad:
        os.makedirs("gpurun_out", exist_ok=True)
        np.savez("gpurun_out/repro_sharded_%d.npz" % seed, data=np.frombuffer(data, dtype=np.uint8), cuts=np.array(cuts), slot=slot)
        print("FAIL case %d: n=%d cuts=%s slot=%d -> status %d records %d, oracle %d %d" % (cases, n, cuts, slot, status, n_records, r.status, r.n_records))
        bnd = [0] + cuts + [n]
        for a_, b_ in zip(bnd[:-1], bnd[1:]):
            if b_ - a_ < 600:
                print("  range [%d, %d): %r" % (a_, b_, data[max(0, a_ - 40): b_ + 40]))
        for sh in shards:
            print("  shard lo=%d hi=%d: status %d records %d head %d tail %d words %s" % (sh.lo, sh.hi, sh.res.status, sh.res.n_records, len(sh.head), len(sh.tail), list(sh.words())))
        break
    cases += 1
print("cases", cases)
