"""Byte-range sharded, host-streamed parsing (BASELINE.json configs[4]): one rank = one GPU = one pinned ring.

The reference's analogue is the gather at the end of Parser::parallel_each (src/lib.rs:553-559) over a pipeline that
a reader thread feeds (src/thread_reader.rs:182-200).  Here the file is cut at arbitrary byte offsets; a rank cannot
wait for the ranks in front of it (they stream for seconds), so it works PHASE-FREE and the ranks talk once, at the
end:

  1. rank r > 0 uploads a few MiB from the start of its range and asks fqh_shard_align for the line phase (newlines in
     front of the range, mod 4) and for the offset R of its first record: the one phase under which the window parses;
  2. it streams [lo + R, hi) through fqh_stream_* exactly like a file of its own (carry zero at lo + R; every record
     validated in the reference's order; histograms added on the way).  What is left behind its last complete record
     is its TAIL; the bytes [lo, lo + R) are its HEAD;
  3. exchange: (status, records, newlines, phase, tail) of every rank.  Rank r checks its phase against the true newline
     count of the ranks in front of it — validity under the true line phase is what the sequential parser computes
      — and parses the STITCH = tail of rank r-1 + its own head as a file of exactly one record
     (the "one-record boundary stitch" of BASELINE.json's north_star);
  4. one all_reduce of [counts, scalars, histograms].

This module is plumbing over the C ABI (binding.py); the exchange itself is done by the caller (torch.distributed in
bench.py, plain lists in the single-process tests) so that it stays testable without a process group."""
import ctypes as C

import numpy as np

from . import binding as B

ALIGN_WINDOW = 4 << 20


class ShardResult:
    __slots__ = ("status", "err_record", "err_offset", "n_records", "n_newlines", "phase", "head", "tail", "bytes", "lo", "hi")

    def summary_words(self):
        """The 8 words a rank contributes to the exchange (the tail bytes travel next to them)."""
        return [self.status, self.n_records, self.n_newlines, self.phase, len(self.head), len(self.tail), self.err_record,
                self.err_offset]


def stream_shard(ctx, read_into, lo, hi, file_len, slot_bytes, n_slots=3, stats=None, d_window=None):
    """Streams bytes [lo, hi) of a file of file_len bytes through a pinned ring on ctx's device.

    read_into(host_addr, file_offset, nbytes) fills pinned memory (a file read, a memcpy, nothing at all for a
    pre-filled benchmark ring).  stats = (lmax, d_qual, d_base, d_scalars) adds every record the rank delivers to
    the histograms.  d_window: device scratch of ALIGN_WINDOW bytes (torch tensor data_ptr or fqh_dev_alloc) for
    fqh_shard_align; only needed when lo > 0.  Returns a ShardResult."""
    res = ShardResult()
    res.lo, res.hi, res.bytes = lo, hi, hi - lo
    res.status, res.err_record, res.err_offset = B.OK, 0, 0
    res.n_records = res.n_newlines = 0
    res.phase, res.head, res.tail = 0, b"", b""
    R = 0
    if lo > 0 and hi > lo:
        w = min(ALIGN_WINDOW, hi - lo)
        hostw = (C.c_uint8 * (w + 1))()
        read_into(C.addressof(hostw), lo - 1, w + 1)  # one byte more in front: is it a newline?
        prev_nl = hostw[0] == 10
        ctx._chk(ctx._L.fqh_memcpy_h2d(ctx._h, d_window, C.addressof(hostw) + 1, w))
        try:
            res.phase, R = ctx.shard_align(d_window, w, prev_nl)
        except B.FqhError as e:
            if e.status not in (B.E_HEADER, B.E_ARG):
                raise
            # the window holds a parse error (or cannot settle the phase): reported as this shard's error at its start
            res.status, res.err_offset = B.E_HEADER, lo
            return res
        res.head = bytes(bytearray(hostw)[1: 1 + R])
        res.n_newlines = res.head.count(b"\n")
    flags = B.STREAM_STATS if stats else 0
    st = B.Stream(ctx, slot_bytes, n_slots, flags)
    try:
        if stats:
            st.set_stats(*stats)
        pos = lo + R
        is_last_shard = hi >= file_len
        done_reading = pos >= hi
        submitted = collected = 0
        end_of_records = pos
        if done_reading:  # nothing but the head
            res.tail = b""
            return res
        while True:
            while not done_reading:
                a = st.acquire()
                if a is None:
                    break
                n = min(a[1], hi - pos)
                read_into(a[0], pos, n)
                pos += n
                done_reading = pos >= hi
                st.submit(n, done_reading and is_last_shard)
                submitted += 1
            if collected == submitted:
                break
            c = st.collect()
            collected += 1
            res.n_records += c.n_records
            rs = np.ctypeslib.as_array(C.cast(c.h_rec_start, C.POINTER(C.c_uint64)), shape=(c.n_records + 1,))
            end_of_records = lo + R + int(rs[c.n_records])  # (the stream's file offsets count from lo + R)
            last_chunk = (c.h_data, c.base_offset, c.data_len, c.lead_len)
            if c.parse_status != B.OK:
                res.status, res.err_record, res.err_offset = c.parse_status, c.err_record, lo + R + c.err_offset
                st.release()
                break
            if collected == submitted and done_reading:
                # what is left behind the last complete record: in pinned memory, in front of / inside the last chunk
                tail_len = hi - end_of_records
                off = end_of_records - (lo + R) - c.base_offset  # relative to h_data (may be negative: in the lead)
                res.tail = C.string_at(c.h_data + off, tail_len) if tail_len else b""
            st.release()
        res.n_newlines += st.carry().nl_count
    finally:
        st.close()
    return res


def check_phases(words):
    """words[r] = summary_words() of rank r, in rank order.  -> list of (rank, message) for ranks whose phase does not
    match the true newline count in front of them (an empty list is the proof that the ranks' local parses add up to
    the sequential one)."""
    bad, nl = [], 0
    for r, w in enumerate(words):
        if r and w[0] == B.OK and (nl & 3) != w[3]:
            bad.append((r, "rank %d parsed at line phase %d, the ranks in front of it hold %d newlines" % (r, w[3], nl)))
        nl += w[2]
    return bad


def stitch(ctx, tail_prev, head, lmax, d_buf, d_qual, d_base, d_scalars):
    """The record that straddles a cut: tail of the previous rank + head of this one, parsed as a file of its own.
    -> (status, n_records); OK means exactly the records of that little file were added to the histograms (one, or none
    when the cut fell on a record boundary).  d_buf: device scratch of at least len(tail_prev) + len(head) + 16 bytes."""
    data = tail_prev + head
    if not data:
        return B.OK, 0
    hb = (C.c_uint8 * len(data)).from_buffer_copy(data)
    ctx._chk(ctx._L.fqh_memcpy_h2d(ctx._h, d_buf, C.addressof(hb), len(data)))
    s, _ = ctx.stats(d_buf, len(data), lmax, d_qual, d_base, d_scalars, is_final=True)
    if s.parse_status == B.OK and s.n_records != 1:
        return B.E_TRUNCATED, s.n_records  # cannot happen for a tail + head of one record; keep the caller honest
    return s.parse_status, s.n_records
