"""The sharded protocol entirely through the C ABI, collectives included (fqh_comm_* / fqh_allgather / fqh_allreduce_u64:
RCCL bound at run time): what a Rust or C++ host without torch would do.  One GPU here, so the communicator has one
rank; the arithmetic of the protocol over several shards is covered by tests/test_gpu_parity.py (4 shards, one GPU) and
tests/test_shard_carry.py (2 ranks, gloo).  The reference's counterpart is the gather of parallel_each's results
(src/lib.rs:553-559)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_shard_protocol_over_the_c_abi_with_rccl(fqref):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as g
    pkg = g.load_package()
    L = pkg.lib()
    dev = torch.device("cuda:0")
    ctx = pkg.Ctx(0)
    n = 330 * 40000
    buf = torch.empty(n + 16, dtype=torch.uint8, device=dev)
    ctx.synth_fill(buf.data_ptr(), 0, n)
    # communicator of this one rank
    uid = C.create_string_buffer(128)
    assert L.fqh_comm_unique_id(uid) == pkg.OK
    comm = C.c_void_p()
    ctx._chk(L.fqh_comm_create(ctx._h, 1, 0, uid, C.byref(comm)))
    # 1) phase-free byte scan of the shard, 2) all_gather of its 7 words, 3) carry, 4) emit with the carry
    nn, ns, back0 = ctx.shard_prescan(buf.data_ptr(), n)
    words = torch.tensor([n, nn, ns] + back0, dtype=torch.int64, device=dev)
    allw = torch.zeros(7, dtype=torch.int64, device=dev)
    ctx._chk(L.fqh_allgather(ctx._h, comm, words.data_ptr(), allw.data_ptr(), 56))
    ctx._chk(L.fqh_sync(ctx._h))
    assert allw.cpu().tolist() == words.cpu().tolist()
    rs = torch.zeros(n // 330 + 2, dtype=torch.int64, device=dev)
    ctx.rescan_launch(True, None, rs.data_ptr(), rs.numel())   # rank 0: no carry in front of it
    s, c, st = ctx.scan_finish()
    assert (s.parse_status, s.n_records) == (pkg.OK, 40000)
    # 5) histograms of the shard, then the all_reduce of [records, errors | scalars | quality | bases]
    tot = torch.zeros(2 + 8 + 150 * 264, dtype=torch.int64, device=dev)
    tot[0] = s.n_records
    ctx.stats(buf.data_ptr(), n, 150, tot[10: 10 + 150 * 256].data_ptr(), tot[10 + 150 * 256:].data_ptr(), tot[2:10].data_ptr())
    torch.cuda.synchronize()
    before = tot.cpu().numpy().copy()
    ctx._chk(L.fqh_allreduce_u64(ctx._h, comm, tot.data_ptr(), tot.numel()))
    ctx._chk(L.fqh_sync(ctx._h))
    after = tot.cpu().numpy()
    assert np.array_equal(before, after)   # (one rank: the sum is the rank's own contribution)
    r, oq, ob, osc = fqref.stats(buf[:n].cpu().numpy(), 150)
    assert np.array_equal(after[2:10].astype(np.uint64), osc)
    assert np.array_equal(after[10: 10 + 150 * 256].astype(np.uint64).reshape(150, 256), oq)
    # the same exchange without the host in between: words -> all-gather -> fold + emit -> sum, one wait at the end
    # (fqh_allgather runs on the context's stream: the three steps are ordered by it)
    W = pkg.SHARD_WORDS
    w1 = torch.zeros(W, dtype=torch.int64, device=dev)
    wall = torch.zeros(W, dtype=torch.int64, device=dev)
    cnt = torch.zeros(2, dtype=torch.int64, device=dev)
    rs.zero_()
    torch.cuda.synchronize()
    ctx.shard_prescan_launch(buf.data_ptr(), n, w1.data_ptr())
    ctx._chk(L.fqh_allgather(ctx._h, comm, w1.data_ptr(), wall.data_ptr(), 8 * W))
    ctx.shard_rescan_launch(True, wall.data_ptr(), 1, 0, rs.data_ptr(), rs.numel(), cnt.data_ptr())
    ctx._chk(L.fqh_allreduce_u64(ctx._h, comm, cnt.data_ptr(), 2))
    s2, c2, st2 = ctx.scan_finish()
    assert (s2.parse_status, s2.n_records, s2.n_newlines) == (pkg.OK, 40000, 160000)
    assert cnt.cpu().tolist() == [40000, 0] and wall.cpu().tolist() == [n, nn, ns] + back0 + [0]
    assert np.array_equal(rs.cpu().numpy()[:40001], np.arange(40001) * 330)
    # the first-error exchange of the sharded modes: element-wise MINIMUM of packed (file offset, rank, kind) keys (ncclMin)
    keys = torch.tensor([pkg.NO_ERROR_KEY - (1 << 64), (12345 << 3) | 2, 7], dtype=torch.int64, device=dev)   # (u64 bit patterns)
    ctx._chk(L.fqh_allreduce_min_u64(ctx._h, comm, keys.data_ptr(), 3))
    ctx._chk(L.fqh_sync(ctx._h))
    assert keys.cpu().tolist() == [-1, (12345 << 3) | 2, 7]
    assert pkg.shard_stream_outcome(pkg.NO_ERROR_KEY, [5, 6, 7]) == (pkg.OK, 18, 0)
    assert pkg.shard_stream_outcome((12345 << 11) | (1 << 3) | 2, [5, 6, 7]) == (pkg.E_LEN_MISMATCH, 11, 12345)
    L.fqh_comm_destroy(comm)
    ctx.close()
