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


def test_stream_fuzzing_bufsize(fqref, env):
    """cfg(fuzzing) BUFSIZE=64 (src/lib.rs:126-127): the too-long replay runs across slot boundaries."""
    torch, pkg = env
    for tag, data in fuzzgen.corpus(4242, 60):
        res = fqref.count(data, bufsize=64)
        status, recs, _ = stream_all(pkg, data, 4096, bufsize=64)
        assert (status, len(recs)) == (res.status, res.n_records), data
