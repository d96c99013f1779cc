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


# ---- the "abi" route of Coll with a stand-in for the library's RCCL binding: a Comm whose three collectives move host bytes
# through gloo (addresses in, addresses out — as the real one takes device addresses).  What is under test is bench.py's side:
# the id made by rank 0 and carried by the process group, the start-up check of the three collectives on known values, the
# verdict every rank takes together, and the fall-back of ALL ranks when one rank's end fails.
FAKE_WORKER = r'''
import ctypes as C, os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
import bench
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cpu")
MODE = os.environ["FAKE_MODE"]

class FqhError(Exception):
    pass

def view(addr, n, dtype=torch.uint8):
    buf = (C.c_uint8 * n).from_address(addr)
    return torch.frombuffer(buf, dtype=torch.uint8).view(dtype)

class Comm:
    @staticmethod
    def unique_id():
        if MODE == "no_rccl":
            raise FqhError("librccl.so could not be loaded")
        return bytes(range(128))
    def __init__(self, ctx, n_ranks, rank_, uid):
        assert bytes(uid) == bytes(range(128)) and n_ranks == world and rank_ == rank
        if MODE == "rank1_fails" and rank == 1:
            raise FqhError("ncclCommInitRank: unhandled system error")
        self.closed = False
    def close(self):
        self.closed = True
    def allgather(self, d_send, d_recv, nbytes):
        out = view(d_recv, nbytes * world)
        dist.all_gather_into_tensor(out, view(d_send, nbytes).clone())
    def allreduce_u64(self, d_buf, n):
        t = view(d_buf, 8 * n, torch.int64)
        if MODE == "wrong_sum" and rank == 0:
            t += 1
        dist.all_reduce(t)
    def allreduce_min_u64(self, d_buf, n):
        t = view(d_buf, 8 * n, torch.int64)      # (the test's keys stay below 2^63: signed order == unsigned order)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
    def sync(self):
        pass

class Pkg:
    pass
pkg = Pkg(); pkg.Comm = Comm; pkg.FqhError = FqhError
coll = bench.Coll(pkg, torch, dist, None, dev, world, rank, "gloo", "abi")
if MODE == "ok":
    assert coll.comm is not None and coll.via_text.startswith("fqh_comm") and "checked on known values" in coll.via_text, coll.via_text
else:
    assert coll.comm is None and coll.via_text.startswith("torch.distributed (gloo)"), coll.via_text
    want = {{"no_rccl": "fqh_comm unavailable", "rank1_fails": "fqh_comm given up: rank 1", "wrong_sum": "fqh_comm given up: rank"}}[MODE]
    assert want in coll.via_text, coll.via_text
# whichever route was agreed on, the collectives work and every rank took the same one
rows = coll.gather_words([rank, 5, (1 << 64) - 1])
assert rows == [[r, 5, (1 << 64) - 1] for r in range(world)], rows
t = torch.tensor([rank + 1, 10], dtype=torch.int64)
coll.sum_dev(t)
assert t.tolist() == [3, 20]
assert coll.min_key(900 + rank) == 900
coll.barrier()
coll.close()
dist.destroy_process_group()
open(os.path.join({out!r}, "ok_%d" % rank), "w").write("ok")
'''


def _run_fake(tmp_path, mode):
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    script = tmp_path / ("fake_%s.py" % mode)
    script.write_text(FAKE_WORKER.format(root=ROOT, out=str(tmp_path)))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FAKE_MODE=mode)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    for r in range(2):
        f = tmp_path / ("ok_%d" % r)
        assert f.exists()
        f.unlink()


def test_coll_abi_route_agreement_over_two_ranks(tmp_path):
    """--comm abi with more than one rank (never possible on this one-GPU box with the real RCCL): the route through a stand-in
    Comm is taken when every rank's end passes its start-up check, and given up by EVERY rank together when rank 0 cannot make an
    id, when one rank's communicator fails to come up, or when a collective returns wrong values."""
    for mode in ("ok", "no_rccl", "rank1_fails", "wrong_sum"):
        _run_fake(tmp_path, mode)
