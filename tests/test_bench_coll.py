"""bench.py's collectives layer (Coll) on CPU, 2 ranks over gloo: the words exchange of the sharded modes, the SUM of u64
counters and the MIN of u64 first-error keys keep every bit of a u64 through torch's int64 tensors (NO_ERROR_KEY = 2^64 - 1
included), on the torch.distributed route that the one-GPU functional mode and a fall-back from the library's own RCCL
binding take.  The gather it stands for: Parser::parallel_each, src/lib.rs:553-559; its error: src/lib.rs:544-547, 561-564."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
import __graft_entry__ as g
import bench
pkg = g.load_package()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cpu")
coll = bench.Coll(pkg, torch, dist, None, dev, world, rank, "gloo", "torch")
assert coll.comm is None and coll.via_text == "torch.distributed (gloo)"
BIG = (1 << 64) - 1
# words: every bit of a u64 survives the exchange
mine = [rank, BIG - rank, 1 << 63, (1 << 63) - 1, 0, 12345678901234567 + rank]
rows = coll.gather_words(mine)
assert rows == [[r, BIG - r, 1 << 63, (1 << 63) - 1, 0, 12345678901234567 + r] for r in range(world)], rows
# SUM of counters, in place
t = torch.tensor([rank + 1, 10, 1 << 40], dtype=torch.int64)
coll.sum_dev(t)
assert t.tolist() == [3, 20, 1 << 41], t.tolist()
# MIN of keys: unsigned order — NO_ERROR_KEY is the largest key, a key with bit 63 set is larger than any without
assert coll.min_key(pkg.NO_ERROR_KEY) == pkg.NO_ERROR_KEY
assert coll.min_key(pkg.NO_ERROR_KEY if rank == 0 else (777 << 11) | (1 << 3) | 2) == (777 << 11) | (1 << 3) | 2
assert coll.min_key((1 << 63) + 5 if rank == 0 else (1 << 63) - 5) == (1 << 63) - 5
assert coll.min_key((1 << 63) + 5 + rank) == (1 << 63) + 5
# device-tensor gather (bytes of any dtype) and the harness's object gather
recv = torch.zeros(world * 3, dtype=torch.uint8)
coll.gather_dev(torch.tensor([rank, 7, 255], dtype=torch.uint8), recv)
assert recv.tolist() == [0, 7, 255, 1, 7, 255]
assert coll.objects({{"r": rank}}) == [{{"r": 0}}, {{"r": 1}}]
coll.barrier()
# a group of one (the N = 1 denominator run by rank 0 alone): no collective is entered
one = bench.Coll(pkg, torch, dist, None, dev, 1, 0, "gloo", "torch", local=True)
assert one.gather_words([BIG, 3]) == [[BIG, 3]] and one.min_key(BIG) == BIG and one.objects(5) == [5]
u = torch.tensor([4], dtype=torch.int64); one.sum_dev(u); assert u.tolist() == [4]
one.barrier()
dist.destroy_process_group()
open(os.path.join({out!r}, "ok_%d" % rank), "w").write("ok")
'''


def test_coll_over_two_gloo_ranks(tmp_path):
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, out=str(tmp_path)))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert (tmp_path / "ok_0").exists() and (tmp_path / "ok_1").exists()
