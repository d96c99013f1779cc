"""tools/exp_long_variant.py LIB — time of the histogram kernels of a cold fqh_stats over 4 GiB of 5 kbp reads with another build
of the library (FQH_LIB_PATH), results NOT checked (knock-out builds count nothing)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["FQH_LIB_PATH"] = os.path.abspath(sys.argv[1])
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda:0")
ctx = pkg.Ctx(0)
read_len = int(os.environ.get("READ_LEN", "5000"))
rng = np.random.default_rng(7)
nrec = 1024
seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), (nrec, read_len))
qual = np.where(rng.random((nrec, read_len)) < 0.8, 126, rng.integers(33, 127, (nrec, read_len))).astype(np.uint8)
if os.environ.get("QUAL_ILLUMINA") == "1":   # '#'..'I' only: inside every kernel's window
    qual = rng.integers(35, 74, (nrec, read_len)).astype(np.uint8)
# READ_LEN_VAR=1: lengths uniform in [READ_LEN / 4, READ_LEN] (the column blocks do not hold equal shares of the bytes any more)
lens = rng.integers(read_len // 4, read_len + 1, nrec) if os.environ.get("READ_LEN_VAR") == "1" else np.full(nrec, read_len)
if os.environ.get("READ_LEN_VAR") == "2":   # a long tail (nanopore-like): log-normal, median READ_LEN / 4, clipped to [100, READ_LEN]
    lens = np.clip(rng.lognormal(np.log(read_len / 4), 0.8, nrec), 100, read_len).astype(np.int64)
block = b"".join(b"@m%06d/ccs\n" % i + seq[i, :lens[i]].tobytes() + b"\n+\n" + qual[i, :lens[i]].tobytes() + b"\n" for i in range(nrec))
reps = (4 << 30) // len(block)
n = reps * len(block)
if os.environ.get("FILE_SORTED") == "1":   # the file's records ordered by length, longest first (every record `reps` times in a row)
    order = np.argsort(-lens, kind="stable")
    one = [b"@m%06d/ccs\n" % i + seq[i, :lens[i]].tobytes() + b"\n+\n" + qual[i, :lens[i]].tobytes() + b"\n" for i in order]
    d = torch.cat([torch.from_numpy(np.frombuffer(r, dtype=np.uint8).copy()).to(dev).repeat(reps) for r in one] + [torch.zeros(16, dtype=torch.uint8, device=dev)])
else:
    d = torch.cat([torch.from_numpy(np.frombuffer(block, dtype=np.uint8).copy()).to(dev).repeat(reps), torch.zeros(16, dtype=torch.uint8, device=dev)])
qh = torch.zeros(read_len * 256, dtype=torch.int64, device=dev); bh = torch.zeros(read_len * 8, dtype=torch.int64, device=dev); sc = torch.zeros(8, dtype=torch.int64, device=dev)
best = 1e9
for _ in range(4):
    ctx.invalidate(); torch.cuda.synchronize()
    ctx.stats(d.data_ptr(), n, read_len, qh.data_ptr(), bh.data_ptr(), sc.data_ptr())
    best = min(best, ctx.timing().stats_ms)
print("%s: histogram kernels %.3f ms (%.0f GB/s)%s" % (os.path.basename(sys.argv[1]), best, n / 1e6 / best, "  rounds=" + os.environ["FQH_LONG_ROUNDS"] if os.environ.get("FQH_LONG_ROUNDS") else ""))
