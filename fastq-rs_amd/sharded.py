"""Byte-range sharded, host-streamed parsing (BASELINE.json configs[4]): ctypes wrapper over the C ABI.

The driver itself — alignment window, ring, head / tail, phase check, one-record stitch, first-error key — lives in the
library (csrc/shard_stream.hip: fqh_shard_stream_run / fqh_shard_stream_finish / fqh_error_key_unpack; the reference's
analogue is Parser::parallel_each, src/lib.rs:509-565, whose parse error is what the whole call returns,
src/lib.rs:544-547, 561-564).  What stays here is plumbing: a Python callable as the read callback, buffers for head and
tail, and the exchange left to the caller (torch.distributed in bench.py, plain lists in the single-process tests, or
fqh_allgather / fqh_allreduce_u64 / fqh_allreduce_min_u64 for hosts without a collective library)."""
import ctypes as C

import numpy as np

from . import binding as B

EDGE_CAP = 2 * B.BUFSIZE   # head and tail of a shard are parts of ONE record: the reference accepts none longer than BUFSIZE


class Shard:
    """One rank's result of fqh_shard_stream_run: res (fqh_shard_result), head and tail bytes, the 8 exchange words."""
    __slots__ = ("res", "head", "tail", "lo", "hi")

    def words(self):
        w = (C.c_uint64 * 8)()
        B.lib().fqh_shard_result_words(C.byref(self.res), C.byref(w))
        return [int(x) for x in w]


def stream_shard(ctx, read_into, lo, hi, file_len, slot_bytes, n_slots=3, stats=None):
    """Streams bytes [lo, hi) of a file of file_len bytes through a pinned ring on ctx's device (fqh_shard_stream_run).
    read_into(host_addr, file_offset, nbytes) fills pinned memory (a file read, a memcpy, nothing at all for a pre-filled
    benchmark ring).  stats = (lmax, d_qual, d_base, d_scalars) adds every record the rank delivers to the histograms."""
    def cb(user, dst, off, n):
        try:
            read_into(dst, off, n)
            return 0
        except Exception:   # (an exception must not cross the C frames)
            return 1
    fn = B.READ_FN(cb)
    sh = Shard()
    sh.res, sh.lo, sh.hi = B.ShardResult(), lo, hi
    head = (C.c_uint8 * EDGE_CAP)()
    tail = (C.c_uint8 * EDGE_CAP)()
    lmax, dq, db, ds = stats if stats else (0, None, None, None)
    ctx._chk(ctx._L.fqh_shard_stream_run(ctx._h, fn, None, lo, hi, file_len, slot_bytes, n_slots, lmax, dq, db, ds,
                                         C.byref(sh.res), C.addressof(head), EDGE_CAP, C.addressof(tail), EDGE_CAP))
    sh.head = bytes(bytearray(head)[: sh.res.head_len])
    sh.tail = bytes(bytearray(tail)[: sh.res.tail_len])
    return sh


def finish(ctx, all_words, all_tails, rank, head, stats=None):
    """After the exchange (all_words[r] = Shard.words() of rank r, all_tails[r] = its tail bytes): phase check, stitch,
    first-error key (fqh_shard_stream_finish) -> (records this rank contributes, key or NO_ERROR_KEY)."""
    n = len(all_words)
    words = np.array(all_words, dtype=np.uint64).reshape(n, 8)
    stride = max(16, max(len(t) for t in all_tails))
    tails = np.zeros((n, stride), dtype=np.uint8)
    for r, t in enumerate(all_tails):
        tails[r, : len(t)] = np.frombuffer(t, dtype=np.uint8)
    hb = (C.c_uint8 * max(1, len(head))).from_buffer_copy(head if head else b"\0")
    out = (C.c_uint64 * 2)()
    lmax, dq, db, ds = stats if stats else (0, None, None, None)
    ctx._chk(ctx._L.fqh_shard_stream_finish(ctx._h, words.ctypes.data, tails.ctypes.data, stride, n, rank, C.addressof(hb), lmax,
                                            dq, db, ds, C.byref(out)))
    return int(out[0]), int(out[1])
