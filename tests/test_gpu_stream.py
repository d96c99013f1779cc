"""GPU tests of the streaming ingest path (fqh_stream_*: pinned ring, side-stream H2D, carry
chaining, host-side contiguity of records across slots) against the oracle's Parser::each."""
import ctypes as C

import numpy as np
import pytest

import fuzzgen

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as g
    pkg = g.load_package()
    return torch, pkg


def stream_all(pkg, data, slot_bytes, n_slots=3, bufsize=None, read_sizes=None):
    """Feeds `data` through a stream; returns (status, records as (head, seq, qual) slices checked via
    the host pointers, boundaries)."""
    ctx = pkg.Ctx(0, bufsize=bufsize)
    st = pkg.Stream(ctx, slot_bytes, n_slots)
    pos, total = 0, len(data)
    recs, bounds = [], [0]
    status = pkg.OK
    submitted = collected = 0
    done_reading = False
    rng = np.random.default_rng(len(data))
    while True:
        # keep the ring full
        while not done_reading:
            a = st.acquire()
            if a is None:
                break
            addr, cap = a
            n = min(cap, total - pos)
            if read_sizes:
                n = min(n, int(rng.integers(1, read_sizes + 1)))
            C.memmove(addr, data[pos: pos + n], n)
            pos += n
            done_reading = pos >= total
            st.submit(n, done_reading)
            submitted += 1
        if collected == submitted:
            break
        c = st.collect()
        collected += 1
        idx = np.ctypeslib.as_array(C.cast(c.h_index, C.POINTER(C.c_uint8)), shape=(max(c.n_records, 1) * 24,)) \
            if c.n_records else np.zeros(0, np.uint8)
        rs = np.ctypeslib.as_array(C.cast(c.h_rec_start, C.POINTER(C.c_uint64)), shape=(c.n_records + 1,)).copy()
        for i in range(c.n_records):
            row = idx[i * 24: (i + 1) * 24]
            start = int(row[:8].view(np.uint64)[0])
            head, seq, sep, qual = (int(x) for x in row[8:].view(np.uint32))
            assert start == int(rs[i])
            assert c.base_offset - start <= c.lead_len or start >= c.base_offset
            raw = C.string_at(c.h_data + (start - c.base_offset), qual + 1)  # contiguous on the host
            assert raw == data[start: start + qual + 1]
            trim = lambda b: b[:-1] if b.endswith(b"\r") else b
            recs.append((trim(raw[1:head]), trim(raw[head + 1: seq]), trim(raw[sep + 1: qual])))
        bounds += [int(x) for x in rs[1:]]
        st.release()
        if c.parse_status != pkg.OK:
            status = c.parse_status
            break
        if c.is_final:
            break
    st.close()
    ctx.close()
    return status, recs, bounds


@pytest.mark.parametrize("seed", range(4))
def test_stream_equals_oracle(fqref, env, seed):
    torch, pkg = env
    rng = np.random.default_rng(300 + seed)
    data = fuzzgen.valid_file(rng, 3000, maxlen=150)
    if seed == 3:
        data = fuzzgen.mutate(rng, data, 1)
    res, idx = fqref.index(data)
    for slot in (4096, 65536, 1 << 20):
        status, recs, bounds = stream_all(pkg, data, slot)
        assert status == res.status
        assert len(recs) == res.n_records
        assert bounds[: res.n_records + 1][:-1] == [int(x) for x in idx[:, 0]]
        for i in (0, len(recs) // 2, len(recs) - 1):
            if recs:
                assert recs[i] == fqref.accessors(data, idx[i])


def test_stream_short_reads_and_tiny_slots(fqref, env):
    torch, pkg = env
    rng = np.random.default_rng(77)
    data = fuzzgen.valid_file(rng, 500, maxlen=60)
    res, idx = fqref.index(data)
    status, recs, bounds = stream_all(pkg, data, 4096, n_slots=2, read_sizes=700)
    assert status == res.status == pkg.OK and len(recs) == res.n_records
    assert [r[1] for r in recs] == [fqref.accessors(data, idx[i])[1] for i in range(res.n_records)]


def test_stream_noted_reads_decide_the_too_long_band(fqref, env):
    """fqh_stream_note_read: a host whose reader comes back with at most CAP bytes per read() notes its reads; the ring then
    replays the reference's Buffer under that reader (src/buffer.rs:51-100 takes what ONE read() returns) and its "too long"
    verdict on records of BUFSIZE - 15 .. BUFSIZE bytes equals the oracle's for the same max_read — which differs from cap to
    cap.  Without notes the reader is a file."""
    torch, pkg = env
    rng = np.random.default_rng(77)
    B = 256

    def sized(total):
        body = total - 6
        s = int(rng.integers(0, body // 2 + 1))
        return b"@" + b"h" * (body - 2 * s) + b"\n" + b"A" * s + b"\n+\n" + b"I" * s + b"\n"

    data = b"".join(sized(int(rng.integers(6, B // 2)) if rng.integers(0, 10) < 6 else int(rng.integers(B - 20, B + 1))) for _ in range(80))
    seen = set()
    for cap in (0, 1, 5, 16, 37, 100, 255, 4096):
        ctx = pkg.Ctx(0, bufsize=B)
        st = pkg.Stream(ctx, 4096, 3)
        pos, nrec, status, done = 0, 0, pkg.OK, False
        submitted = collected = 0
        while True:
            while not done:
                a = st.acquire()
                if a is None:
                    break
                addr, room = a
                n = 0
                while n < room and pos < len(data):      # the host fills the slot read by read
                    asked = room - n
                    got = min(asked, len(data) - pos, cap or asked)
                    C.memmove(addr + n, data[pos: pos + got], got)
                    if cap:
                        st.note_read(got, asked)
                    n += got
                    pos += got
                done = pos >= len(data)
                st.submit(n, done)
                submitted += 1
            if collected == submitted:
                break
            c = st.collect()
            collected += 1
            nrec += c.n_records
            st.release()
            if c.parse_status != pkg.OK or c.is_final:
                status = c.parse_status
                break
        st.close()
        ctx.close()
        r = fqref.count(data, bufsize=B, max_read=cap)
        assert (status, nrec) == (r.status, r.n_records), (cap, status, nrec, r.status, r.n_records)
        seen.add((r.status, r.n_records))
    assert len(seen) > 1, seen   # (the caps DID decide)
    # notes that begin after the first submit are refused
    ctx = pkg.Ctx(0, bufsize=B)
    st = pkg.Stream(ctx, 4096, 3)
    st.acquire()
    st.submit(0, False)
    with pytest.raises(pkg.FqhError):
        st.note_read(10, 20)
    st.close()
    ctx.close()


def test_stream_too_long_and_truncation(fqref, env):
    torch, pkg = env
    rng = np.random.default_rng(5)
    B = fqref.BUFSIZE
    pre = fuzzgen.valid_file(rng, 50)
    for L in (B - 16, B - 3, B, B + 1, B + 5000):
        data = pre + b"@" + b"h" * (L - 9) + b"\nA\n+\nB\n" + fuzzgen.valid_file(rng, 20)
        res = fqref.count(data)
        status, recs, bounds = stream_all(pkg, data, 1 << 16)
        assert (status, len(recs)) == (res.status, res.n_records), L
    data = pre + b"@x\nACGT\n+\nII"
    res = fqref.count(data)
    status, recs, _ = stream_all(pkg, data, 8192)
    assert (status, len(recs)) == (res.status, res.n_records) == (pkg.E_TRUNCATED, 50)


def test_stream_dense_records_with_trailing_error(fqref, env):
    """Records of 6-8 bytes (more of them per slot than the per-slot boundary arrays hold at first) AND a parse error in the
    same chunk: the boundaries of every record before the error come back, none past the arrays (the capacity
    check used to be skipped whenever the chunk carried a parse error)."""
    torch, pkg = env
    rng = np.random.default_rng(9)
    recs = [b"@\n" + (b"A" * k) + b"\n+\n" + (b"I" * k) + b"\n" for k in rng.integers(0, 2, 5000)]
    for tail in (b"@x\nAC", b"@x\nAC\n+\nI\n", b"x\n"):
        data = b"".join(recs) + tail
        res, idx = fqref.index(data)
        assert res.status != pkg.OK and res.n_records == 5000
        for slot in (4096, 1 << 16):
            status, got, bounds = stream_all(pkg, data, slot)
            assert (status, len(got)) == (res.status, res.n_records), (tail, slot)
            assert bounds[: res.n_records] == [int(x) for x in idx[:, 0]]


def test_stream_fuzzing_bufsize(fqref, env):
    """cfg(fuzzing) BUFSIZE=64 (src/lib.rs:126-127): the too-long replay runs across slot boundaries."""
    torch, pkg = env
    for tag, data in fuzzgen.corpus(4242, 60):
        res = fqref.count(data, bufsize=64)
        status, recs, _ = stream_all(pkg, data, 4096, bufsize=64)
        assert (status, len(recs)) == (res.status, res.n_records), data


def stream_stats(torch, pkg, data, slot_bytes, lmax, n_slots=3, read_sizes=None, seed=0, routes=None):
    """Feeds `data` through a FQH_STREAM_STATS stream; returns (status, n_records, qual, base, scalars)."""
    dev = torch.device("cuda:0")
    ctx = pkg.Ctx(0)
    st = pkg.Stream(ctx, slot_bytes, n_slots, pkg.STREAM_STATS)
    qh = torch.zeros(lmax * 256, dtype=torch.int64, device=dev)
    bh = torch.zeros(lmax * 8, dtype=torch.int64, device=dev)
    sc = torch.zeros(8, dtype=torch.int64, device=dev)
    st.set_stats(lmax, qh.data_ptr(), bh.data_ptr(), sc.data_ptr())
    pos, total = 0, len(data)
    status, nrec = pkg.OK, 0
    submitted = collected = 0
    done_reading = False
    rng = np.random.default_rng(seed)
    while True:
        while not done_reading:
            a = st.acquire()
            if a is None:
                break
            addr, cap = a
            n = min(cap, total - pos)
            if read_sizes:
                n = min(n, int(rng.integers(1, read_sizes + 1)))
            C.memmove(addr, data[pos: pos + n], n)
            pos += n
            done_reading = pos >= total
            st.submit(n, done_reading)
            submitted += 1
        if collected == submitted:
            break
        c = st.collect()
        collected += 1
        nrec += c.n_records
        if routes is not None:
            routes.append((c.data_len, bool(ctx.last_scan_fast())))
        st.release()
        if c.parse_status != pkg.OK:
            status = c.parse_status
            break
        if c.is_final:
            break
    torch.cuda.synchronize()
    out = (status, nrec, qh.cpu().numpy().astype(np.uint64).reshape(lmax, 256),
           bh.cpu().numpy().astype(np.uint64).reshape(lmax, 8), sc.cpu().numpy().astype(np.uint64))
    st.close()
    ctx.close()
    return out


@pytest.mark.parametrize("seed", range(5))
def test_streamed_stats_equal_whole_file_oracle(fqref, env, seed):
    """FQH_STREAM_STATS: histograms accumulated chunk by chunk == the oracle's on the whole file, for
    slots far smaller than the file (every slot boundary cuts a record; some records span several
    slots), ragged reads, CRLF and files that end in an error (records before it still count)."""
    torch, pkg = env
    rng = np.random.default_rng(7000 + seed)
    if seed == 0:
        data, slot, reads = fuzzgen.valid_file(rng, 3000, maxlen=150), 4096, None
    elif seed == 1:
        data, slot, reads = fuzzgen.valid_file(rng, 2000, maxlen=300, crlf=True), 8192, 3000
    elif seed == 2:   # long records against small slots: a record spans several chunks
        data, slot, reads = fuzzgen.valid_file(rng, 60, seqlen=9000), 4096, None
    elif seed == 3:
        data, slot, reads = fuzzgen.mutate(rng, fuzzgen.valid_file(rng, 2500, maxlen=150), 1), 16384, 5000
    else:
        data, slot, reads = fuzzgen.valid_file(rng, 4000, maxlen=150)[:-9], 65536, None
    for lmax in (150, 64):
        r, qh, bh, sc = fqref.stats(data, lmax)
        status, nrec, gq, gb, gs = stream_stats(torch, pkg, data, slot, lmax, read_sizes=reads, seed=seed)
        assert (status, nrec) == (r.status, r.n_records), (seed, lmax)
        assert np.array_equal(gs, sc), (seed, lmax, gs, sc)
        assert np.array_equal(gq, qh) and np.array_equal(gb, bh), (seed, lmax)


def test_stats_lead_chunks_add_up(fqref, env):
    """fqh_stats_launch_lead on hand-made chunks of one device buffer (the bytes in front of a chunk
    are simply the rest of the file): the chunked calls add up to the whole-file oracle; without the
    lead every cut loses exactly the record it goes through."""
    torch, pkg = env
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(99)
    data = fuzzgen.valid_file(rng, 5000, maxlen=200)
    n = len(data)
    d = torch.empty(n + 64, dtype=torch.uint8, device=dev)
    d[:n].copy_(torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()))
    cuts = [0] + sorted(int(x) // 16 * 16 for x in rng.integers(1, n, 7)) + [n]   # 16-byte aligned chunk starts
    lmax = 200
    r, qh, bh, sc = fqref.stats(data, lmax)
    for use_lead in (True, False):
        ctx = pkg.Ctx(0)
        gq = torch.zeros(lmax * 256, dtype=torch.int64, device=dev)
        gb = torch.zeros(lmax * 8, dtype=torch.int64, device=dev)
        gs = torch.zeros(8, dtype=torch.int64, device=dev)
        carry, total, lost = None, 0, 0
        for a, b in zip(cuts[:-1], cuts[1:]):
            if a == b:
                continue
            if use_lead:
                ctx.stats_launch_lead(d.data_ptr() + a, b - a, a, lmax, gq.data_ptr(), gb.data_ptr(), gs.data_ptr(),
                                      is_final=(b == n), carry=carry)
            else:
                ctx.stats_launch(d.data_ptr() + a, b - a, lmax, gq.data_ptr(), gb.data_ptr(), gs.data_ptr(),
                                 is_final=(b == n), carry=carry)
            s, c = ctx.stats_finish()
            assert s.parse_status == pkg.OK
            total += s.n_records
            if carry is not None and carry.back[carry.nl_count & 3] > 0 and s.n_records:
                lost += 1
            carry = c
        assert total == r.n_records
        got = gs.cpu().numpy().astype(np.uint64)
        if use_lead:
            assert np.array_equal(got, sc)
            assert np.array_equal(gq.cpu().numpy().astype(np.uint64).reshape(lmax, 256), qh)
            assert np.array_equal(gb.cpu().numpy().astype(np.uint64).reshape(lmax, 8), bh)
        else:
            assert got[0] == sc[0] - lost and lost > 0
        ctx.close()


def _clean_reads(rng, nrec, L, crlf=False):
    alph = np.frombuffer(b"ACGTN", dtype=np.uint8)
    e = b"\r\n" if crlf else b"\n"
    out = []
    for i in range(nrec):
        n = L if isinstance(L, int) else int(rng.integers(L[0], L[1] + 1))
        seq = rng.choice(alph, n, p=[.2475, .2475, .2475, .2475, .01]).tobytes()
        qual = rng.integers(33, 75, n).astype(np.uint8).tobytes()
        out.append(b"@read%d 1:N:0" % i + e + seq + e + b"+" + e + qual + e)
    return b"".join(out)


@pytest.mark.parametrize("shape", ["fixed150", "ragged", "crlf"])
def test_streamed_stats_take_the_single_pass(fqref, env, shape):
    """FQH_STREAM_STATS on ordinary reads: every slot is scanned AND counted in one read of its bytes (k_scan_stats with the
    slot's carry; k_stats_edge for the record that straddles into the slot and for the partial one at its end), the way the
    reference touches a record once whatever the buffer (src/lib.rs:226-237, src/records.rs:83-90).  Values == oracle; the
    route is pinned: the fast path stood for every slot but (possibly) the short last one."""
    torch, pkg = env
    rng = np.random.default_rng(4242)
    data = _clean_reads(rng, 30000, 150 if shape == "fixed150" else (20, 150), crlf=(shape == "crlf"))
    lmax = 150
    r, qh, bh, sc = fqref.stats(data, lmax)
    for slot in (1 << 20, 300000 // 16 * 16):
        routes = []
        status, nrec, gq, gb, gs = stream_stats(torch, pkg, data, slot, lmax, routes=routes)
        assert (status, nrec) == (r.status, r.n_records) == (pkg.OK, 30000)
        assert np.array_equal(gs, sc), (gs, sc)
        assert np.array_equal(gq, qh) and np.array_equal(gb, bh)
        full = [fast for n, fast in routes if n == slot]
        assert len(full) >= 3 and all(full), routes


def test_chunked_stats_take_the_single_pass(fqref, env):
    """fqh_stats_launch_lead over consecutive chunks of one device buffer, cut anywhere (16-byte aligned): each chunk is ONE
    read of its bytes (single pass kept) and the chunked calls add up to the whole-file oracle."""
    torch, pkg = env
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(515)
    data = _clean_reads(rng, 40000, (100, 150))
    n = len(data)
    d = torch.empty(n + 64, dtype=torch.uint8, device=dev)
    d[:n].copy_(torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()))
    lmax = 150
    r, qh, bh, sc = fqref.stats(data, lmax)
    cuts = [0] + sorted(int(x) // 16 * 16 for x in rng.integers(1 << 20, n - (1 << 20), 5)) + [n]
    ctx = pkg.Ctx(0)
    gq = torch.zeros(lmax * 256, dtype=torch.int64, device=dev)
    gb = torch.zeros(lmax * 8, dtype=torch.int64, device=dev)
    gs = torch.zeros(8, dtype=torch.int64, device=dev)
    carry, total = None, 0
    for a, b in zip(cuts[:-1], cuts[1:]):
        if a == b:
            continue
        ctx.stats_launch_lead(d.data_ptr() + a, b - a, a, lmax, gq.data_ptr(), gb.data_ptr(), gs.data_ptr(), is_final=(b == n), carry=carry)
        s, c = ctx.stats_finish()
        assert s.parse_status == pkg.OK, (a, b)
        assert ctx.last_scan_fast() or b - a < (1 << 18), (a, b)   # (a chunk of a few tiles may not settle an alignment)
        total += s.n_records
        carry = c
    ctx.close()
    assert total == r.n_records
    assert np.array_equal(gs.cpu().numpy().astype(np.uint64), sc)
    assert np.array_equal(gq.cpu().numpy().astype(np.uint64).reshape(lmax, 256), qh)
    assert np.array_equal(gb.cpu().numpy().astype(np.uint64).reshape(lmax, 8), bh)


def test_filter_between_collects_takes_a_second_context(fqref, env):
    """Between two collects the context belongs to the stream (a collect may already have enqueued the next slot's scan on it,
    include/fastq_hip.h): a statistics / filter call on the SAME context says so (FQH_E_ARG, no crash, the stream goes on), the
    same call on a second context of the device works on the collected chunk's device data, and a two-slot ring still delivers
    the oracle's records (the refill of a slot waits for the event recorded in front of the look-ahead scan)."""
    torch, pkg = env
    rng = np.random.default_rng(77)
    data = fuzzgen.valid_file(rng, 30000, maxlen=100, crlf=False)
    r, off = fqref.offsets(data)
    ctx, side = pkg.Ctx(0), pkg.Ctx(0)
    st = pkg.Stream(ctx, 1 << 18, 2, 0)          # two slots, no index: the look-ahead launch is taken whenever a slot is waiting
    pos, n_rec, refused, flagged = 0, 0, 0, 0
    sub = col = 0
    done = False
    while True:
        while not done:
            a = st.acquire()
            if a is None:
                break
            n = min(a[1], len(data) - pos)
            C.memmove(a[0], data[pos: pos + n], n)
            pos += n
            done = pos >= len(data)
            st.submit(n, done)
            sub += 1
        if col == sub:
            break
        c = st.collect()
        col += 1
        assert c.parse_status == pkg.OK
        if c.n_records:
            dummy = torch.zeros(64, dtype=torch.uint8, device="cuda")
            try:                                  # the stream's own context: refused while a launch of the stream is pending
                ctx.scan(dummy.data_ptr(), 16)
            except pkg.FqhError as e:
                assert e.status == pkg.E_ARG
                refused += 1
            s2 = side.scan(c.d_data, c.data_len, bool(c.is_final))[0]   # a second context scans the chunk's device twin
            assert s2.n_newlines == data[c.base_offset: c.base_offset + c.data_len].count(b"\n")
            flagged += 1
        n_rec += c.n_records
        st.release()
        if c.is_final:
            break
    st.close()
    ctx.close()
    side.close()
    assert n_rec == r.n_records == 30000 and flagged >= 5 and refused >= 1


def stream_external(torch, pkg, data, slot_bytes, lmax, mode, seed, n_slots=3, registered=True):
    """As stream_stats, but the slots' bytes come straight from the caller's memory (fqh_stream_submit_external).
    mode "external": a ring of FQH_STREAM_EXTERNAL (no pinned data slots); "mixed": an ordinary ring whose slots are fed either
    way at random.  Pieces of random size (a few bytes to a slot), so records straddle chunks of either kind.
    -> (status, n_records, qual, base, scalars, boundaries)"""
    dev = torch.device("cuda:0")
    ctx = pkg.Ctx(0)
    flags = pkg.STREAM_STATS | (pkg.STREAM_EXTERNAL if mode == "external" else 0)
    st = pkg.Stream(ctx, slot_bytes, n_slots, flags)
    qh = torch.zeros(lmax * 256, dtype=torch.int64, device=dev)
    bh = torch.zeros(lmax * 8, dtype=torch.int64, device=dev)
    sc = torch.zeros(8, dtype=torch.int64, device=dev)
    st.set_stats(lmax, qh.data_ptr(), bh.data_ptr(), sc.data_ptr())
    src = np.frombuffer(data, dtype=np.uint8).copy() if len(data) else np.zeros(1, np.uint8)
    if registered:
        ctx.host_register(src.ctypes.data, max(1, len(data)))
    if mode == "external":
        assert st.acquire_status() == pkg.E_ARG     # no pinned data area to hand out
    pos, total = 0, len(data)
    status, nrec, bounds = pkg.OK, 0, [0]
    in_flight = 0
    done_reading = False
    rng = np.random.default_rng(seed)
    while True:
        while not done_reading and in_flight < n_slots:
            n = min(slot_bytes, total - pos, int(rng.integers(1, slot_bytes + 1)) if rng.integers(0, 3) else slot_bytes)
            last = pos + n >= total
            if mode == "mixed" and rng.integers(0, 2):
                a = st.acquire()
                assert a is not None
                C.memmove(a[0], src.ctypes.data + pos, n)
                st.submit(n, last)
            else:
                assert st.submit_external(src.ctypes.data + pos, n, last)
            pos += n
            done_reading = last
            in_flight += 1
        if mode == "external" and not done_reading:
            assert not st.submit_external(src.ctypes.data + pos, 1, False)   # the ring is full: FQH_E_CAPACITY, nothing taken
        if in_flight == 0:
            break
        c = st.collect()
        in_flight -= 1
        nrec += c.n_records
        rs = np.ctypeslib.as_array(C.cast(c.h_rec_start, C.POINTER(C.c_uint64)), shape=(c.n_records + 1,)).copy()
        bounds += [int(x) for x in rs[1:]]
        st.release()
        if c.parse_status != pkg.OK:
            status = c.parse_status
            break
        if c.is_final:
            break
    torch.cuda.synchronize()
    out = (status, nrec, qh.cpu().numpy().astype(np.uint64).reshape(lmax, 256),
           bh.cpu().numpy().astype(np.uint64).reshape(lmax, 8), sc.cpu().numpy().astype(np.uint64), bounds)
    st.close()
    if registered:
        ctx.host_unregister(src.ctypes.data)
    ctx.close()
    return out


@pytest.mark.parametrize("mode", ["external", "mixed"])
@pytest.mark.parametrize("seed", range(4))
def test_stream_external_source_equals_oracle(fqref, env, mode, seed):
    """fqh_stream_submit_external: the DMA engine reads the caller's registered memory, no staging copy (the copy
    src/thread_reader.rs:90-97 makes).  Boundaries, status and histograms are the oracle's whatever the pieces; records in
    progress cross chunks of the caller's memory and ring slots alike (seed 3: a mutated file; seed 2: pageable memory)."""
    torch, pkg = env
    rng = np.random.default_rng(900 + seed)
    data = fuzzgen.valid_file(rng, 4000, maxlen=150)
    if seed == 3:
        data = fuzzgen.mutate(rng, data, 1)
    res, idx = fqref.index(data)
    r, oq, ob, osc = fqref.stats(data, 150)
    for slot in (4096, 1 << 16, 1 << 20):
        status, nrec, q, b, sc, bounds = stream_external(torch, pkg, data, slot, 150, mode, seed * 7 + slot, registered=seed != 2)
        assert (status, nrec) == (res.status, res.n_records)
        assert bounds[: res.n_records + 1][:-1] == [int(x) for x in idx[:, 0]]
        if res.status == pkg.OK:
            assert np.array_equal(q, oq) and np.array_equal(b, ob) and np.array_equal(sc, osc)


def test_keep_ring_hands_the_pinned_slots_to_the_next_ring(fqref, env):
    """FQH_OPT_KEEP_RING: a destroyed ring's pinned slots stay with the context; the next ring of the same geometry gets the
    same memory (nothing is pinned again), a ring of another geometry gets its own, and results do not care."""
    torch, pkg = env
    rng = np.random.default_rng(5)
    data = fuzzgen.valid_file(rng, 2000, maxlen=100)
    res = fqref.count(data)
    ctx = pkg.Ctx(0)
    ctx.set_keep_ring(True)

    def run(slot_bytes):
        st = pkg.Stream(ctx, slot_bytes, 3, 0)
        addrs, pos, n_rec, done = [], 0, 0, False
        sub = col = 0
        while True:
            while not done:
                a = st.acquire()
                if a is None:
                    break
                addrs.append(a[0])
                n = min(a[1], len(data) - pos)
                C.memmove(a[0], data[pos: pos + n], n)
                pos += n
                done = pos >= len(data)
                st.submit(n, done)
                sub += 1
            if col == sub:
                break
            c = st.collect()
            col += 1
            n_rec += c.n_records
            st.release()
            assert c.parse_status == pkg.OK
        st.close()
        return set(addrs[:3]), n_rec

    a1, n1 = run(1 << 16)
    a2, n2 = run(1 << 16)
    a3, n3 = run(1 << 15)       # another geometry: memory of its own (and the parked ring stays: it is the bigger one)
    a4, n4 = run(1 << 16)
    assert n1 == n2 == n3 == n4 == res.n_records
    assert a1 == a2 == a4 and not (a1 & a3)
    ctx.set_keep_ring(False)    # frees what is parked
    ctx.close()


def test_stream_noted_reads_too_long_names_the_record(fqref, env):
    """fqh_stream_note_read, files of mostly band-sized records (BUFSIZE - 17 .. BUFSIZE) in slots whose ends fall anywhere: the chunk
    that reports "Fastq record is too long" names the record Parser::each stops at under the same reader (the oracle's max_read) —
    err_record, err_offset — and delivers exactly the records in front of it.  ADVICE r5 asked whether the replay, which stops
    where the reference's reader would block, can judge a record only AFTER its chunk has handed it out; it cannot: the
    reference's verdict falls when its buffer is full and the record still open, i.e. before the record's last byte is read
    (src/lib.rs:276-283, src/buffer.rs:51-72), so every byte the verdict needs lies in the chunk the record ends in.  300+
    too-long cases here, none judged late (the ring still reports such a case correctly, should one exist: stream.hip)."""
    torch, pkg = env
    B, slot = 256, 4096
    late = checked = 0
    for seed in range(400):
        rng = np.random.default_rng(7000 + seed)

        def sized(total):
            body = total - 6
            s = int(rng.integers(0, body // 2 + 1))
            return b"@" + b"h" * (body - 2 * s) + b"\n" + b"A" * s + b"\n+\n" + b"I" * s + b"\n"

        data = b"".join(sized(int(rng.integers(B - 17, B + 1))) if rng.integers(0, 3) else sized(int(rng.integers(6, 60))) for _ in range(70))
        cap = int(rng.choice([5, 16, 37, 100, 255]))
        r = fqref.count(data, bufsize=B, max_read=cap)
        if r.status != pkg.E_TOO_LONG:
            continue
        _, idx = fqref.index(data, bufsize=1 << 20)
        ctx = pkg.Ctx(0, bufsize=B)
        st = pkg.Stream(ctx, slot, 3)
        pos, delivered, done = 0, 0, False
        submitted = collected = 0
        verdict = None
        while verdict is None:
            while not done:
                a = st.acquire()
                if a is None:
                    break
                addr, room = a
                n = 0
                while n < room and pos < len(data):
                    asked = room - n
                    got = min(asked, len(data) - pos, cap)
                    C.memmove(addr + n, data[pos: pos + got], got)
                    st.note_read(got, asked)
                    n += got
                    pos += got
                done = pos >= len(data)
                st.submit(n, done)
                submitted += 1
            if collected == submitted:
                break
            c = st.collect()
            collected += 1
            if c.parse_status != pkg.OK:
                verdict = (c.parse_status, int(c.n_records), int(c.err_record), int(c.err_offset), delivered)
            delivered += c.n_records
            st.release()
            if c.is_final:
                break
        st.close()
        ctx.close()
        assert verdict is not None and verdict[0] == pkg.E_TOO_LONG, (seed, cap, verdict, r.n_records)
        status, n_here, err_record, err_offset, before = verdict
        checked += 1
        assert err_record == r.n_records, (seed, cap, verdict, r.n_records)          # the record the reference stops at ...
        assert err_offset == int(idx[r.n_records][0]), (seed, cap, verdict)            # ... and where it begins
        if err_record < before:                                                         # judged one chunk late
            late += 1
            assert n_here == 0
        else:
            assert before + n_here == r.n_records
    assert checked >= 100, checked
    assert late == 0, late
