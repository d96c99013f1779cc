"""Host logic of byte-range sharding on CPU: fqh_carry_combine folds per-shard
zero-carry summaries into each shard's true carry-in.  Checked against the carry computed directly
from the file prefix, single-process and across 2 ranks over gloo (the N > 1 path of bench.py:
all_gather of 7 words per rank, fold in rank order); the 2-rank exchange is driven by the words the byte-scan kernel produced
on an MI355X (tests/golden/shard_words_gpu.json)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import fuzzgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def truth_carry(data, x):
    a = np.frombuffer(data, dtype=np.uint8)[:x]
    nl = np.flatnonzero(a == 10)
    starts = np.concatenate(([0], nl + 1))[::-1]
    back = [int(x - starts[i]) if i < starts.size else int(x) for i in range(4)]
    return x, int(nl.size), back


def shard_summary(data, a, b):
    """What fqh_scan(in=NULL, is_final=0) reports for shard [a, b): newlines, line starts at shard
    offsets 1..len, distances from the shard end to the most recent of them."""
    s = np.frombuffer(data, dtype=np.uint8)[a:b]
    nl = np.flatnonzero(s == 10)
    starts = (nl + 1)[::-1]
    n = b - a
    back0 = [int(n - starts[i]) if i < starts.size else n for i in range(4)]
    return n, int(nl.size), int(nl.size), back0


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g
    p = g.load_package()
    if not os.path.exists(p.LIB_PATH):
        g.build()
    return p


@pytest.mark.parametrize("seed", range(5))
def test_carry_combine_matches_prefix_truth(pkg, seed):
    rng = np.random.default_rng(seed)
    data = fuzzgen.valid_file(rng, 200, maxlen=50) if seed < 4 else b"x" * 500 + b"\n" + b"y" * 300
    for trial in range(20):
        k = int(rng.integers(1, 9))
        cuts = [0] + sorted(int(x) for x in rng.integers(0, len(data) + 1, k)) + [len(data)]
        carry = None
        for a, b in zip(cuts[:-1], cuts[1:]):
            carry = pkg.carry_combine(carry, *shard_summary(data, a, b))
            base, nlc, back = truth_carry(data, b)
            assert (carry.base_offset, carry.nl_count, list(carry.back)) == (base, nlc, back)


WORKER = r'''
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
import __graft_entry__ as g, fuzzgen
from test_shard_carry import shard_summary, truth_carry
pkg = g.load_package()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
# the words are the byte-scan KERNEL's (fqh_shard_prescan on an MI355X, tests/golden/shard_words_gpu.json), the file is
# regenerated from its seed; numpy says what the words must be
import json
from test_shard_fixture import fixture_file
fx = json.load(open(os.path.join({root!r}, "tests", "golden", "shard_words_gpu.json")))["hbm_ranks2"]
data = fixture_file(fx["file"])
cut = [0] + fx["cuts"] + [len(data)]
row = fx["prescan_words"][rank]
mine = (row[0], row[1], row[2], row[3:7])
assert list(shard_summary(data, cut[rank], cut[rank + 1])[:3]) == row[:3] and shard_summary(data, cut[rank], cut[rank + 1])[3] == row[3:7]
t = torch.tensor([mine[0], mine[1], mine[2]] + mine[3], dtype=torch.int64)
outs = [torch.zeros(7, dtype=torch.int64) for _ in range(world)]
dist.all_gather(outs, t)
carry = None
for r in range(rank):
    row = outs[r].tolist()
    carry = pkg.carry_combine(carry, row[0], row[1], row[2], row[3:7])
base, nlc, back = truth_carry(data, cut[rank])
got = (0, 0, [0, 0, 0, 0]) if carry is None else (carry.base_offset, carry.nl_count, list(carry.back))
assert got == (base, nlc, back), (rank, got, (base, nlc, back))
# the final "all-reduce of counts" of bench.py
c = torch.tensor([mine[1]], dtype=torch.int64)
dist.all_reduce(c)
assert int(c) == data.count(b"\n")
dist.destroy_process_group()
open(os.path.join({out!r}, "ok_%d" % rank), "w").write("ok")
'''


def test_two_rank_gloo_carry_exchange(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, out=str(tmp_path)))
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", port, str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert (tmp_path / "ok_0").exists() and (tmp_path / "ok_1").exists(), out.stdout + out.stderr
