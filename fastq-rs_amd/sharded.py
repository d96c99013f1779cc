"""Byte-range sharded, host-streamed parsing (BASELINE.json configs[4]): ctypes wrapper over the C ABI.

The driver itself — alignment window, ring, true-phase check after the exchange, the parse of the gaps between the ranks'
records, first-error key — lives in the library (csrc/shard_stream.hip: fqh_shard_stream_run / fqh_shard_stream_finish /
fqh_shard_stream_outcome; the reference's analogue is Parser::parallel_each, src/lib.rs:509-565, whose parse error is what the
whole call returns, src/lib.rs:544-547, 561-564).  What stays here is plumbing: a Python callable as the read callback and
the exchange left to the caller (torch.distributed in bench.py, plain lists in the single-process tests, or fqh_allgather /
fqh_allreduce_u64 / fqh_allreduce_min_u64 for hosts without a collective library)."""
import ctypes as C

import numpy as np

from . import binding as B


def _callback(read_into):
    def cb(user, dst, off, n):
        try:
            read_into(dst, off, n)
            return 0
        except Exception:   # (an exception must not cross the C frames)
            return 1
    return B.READ_FN(cb)


class Shard:
    """One rank's result of fqh_shard_stream_run: res (fqh_shard_result) or, if the run failed, the status it failed with;
    words() is what the rank sends into the exchange either way (a rank that failed must still take part)."""
    __slots__ = ("res", "lo", "hi", "failed")

    def words(self):
        if self.failed:
            return B.shard_failed_words(self.failed, self.lo, self.hi)
        w = (C.c_uint64 * B.SHARD_STREAM_WORDS)()
        B.lib().fqh_shard_result_words(C.byref(self.res), self.lo, self.hi, C.byref(w))
        return [int(x) for x in w]


def _map_callback(map_at):
    def cb(user, off, want, avail):
        try:
            r = map_at(off, want)
            if not r:
                return None
            avail[0] = int(r[1])
            return int(r[0])
        except Exception:
            return None
    return B.MAP_FN(cb)


def stream_shard(ctx, read_into, lo, hi, file_len, slot_bytes, n_slots=3, stats=None, map_at=None):
    """Streams bytes [lo, hi) of a file of file_len bytes through a pinned ring on ctx's device (fqh_shard_stream_run).
    read_into(host_addr, file_offset, nbytes) fills pinned memory (a file read, a memcpy, nothing at all for a pre-filled
    benchmark ring).  stats = (lmax, d_qual, d_base, d_scalars) adds every record the rank delivers to the histograms (the
    rank's own arrays, zeroed).  A failure of the run (the callback raised, a device error) is kept in Shard.failed, not raised:
    the other ranks wait in the exchange.  map_at(file_offset, want) -> (host_addr, avail) or None: the streamed bytes are taken in
    place from page-locked host memory instead (fqh_shard_stream_run_mapped; read_into still serves the alignment window)."""
    fn = _callback(read_into)
    sh = Shard()
    sh.res, sh.lo, sh.hi, sh.failed = B.ShardResult(), lo, hi, 0
    lmax, dq, db, ds = stats if stats else (0, None, None, None)
    if map_at is not None:
        mfn = _map_callback(map_at)
        st = ctx._L.fqh_shard_stream_run_mapped(ctx._h, fn, mfn, None, lo, hi, file_len, slot_bytes, n_slots, lmax, dq, db, ds,
                                                C.byref(sh.res))
    else:
        st = ctx._L.fqh_shard_stream_run(ctx._h, fn, None, lo, hi, file_len, slot_bytes, n_slots, lmax, dq, db, ds, C.byref(sh.res))
    if st != B.OK:
        sh.failed = st
    return sh


def finish(ctx, read_into, file_len, all_words, rank, slot_bytes, n_slots=3, stats=None):
    """After the exchange (all_words[r] = Shard.words() of rank r): true-phase check, the parse of the gap in front of this rank
    (or behind the last rank that parsed under the true phase), first-error key (fqh_shard_stream_finish) -> (records this rank
    contributes, key or NO_ERROR_KEY).  A failure here becomes a failure key: the reductions that follow need every rank."""
    n = len(all_words)
    words = np.array(all_words, dtype=np.uint64).reshape(n, B.SHARD_STREAM_WORDS)
    fn = _callback(read_into)
    out = (C.c_uint64 * 2)()
    lmax, dq, db, ds = stats if stats else (0, None, None, None)
    st = ctx._L.fqh_shard_stream_finish(ctx._h, fn, None, file_len, words.ctypes.data, n, rank, slot_bytes, n_slots, lmax, dq, db, ds,
                                        C.byref(out))
    if st != B.OK:
        return 0, B.shard_failure_key(rank, int(words[rank, 8]), st)
    return int(out[0]), int(out[1])


def outcome(records_per_rank, min_key):
    """-> (status, n_records, err_offset): Parser::each's result over the whole file (fqh_shard_stream_outcome)."""
    return B.shard_stream_outcome(min_key, records_per_rank)
