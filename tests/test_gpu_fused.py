"""GPU parity tests of the single-pass route (k_scan_stats, fqh_stats / fqh_scan_stats on a whole file): one read of
the input gives record offsets AND histograms, the way the reference's Parser::each hands each record to the closure
that reads seq()/qual() (src/lib.rs:226-237, src/records.rs:75-90).  Every case is compared bit-for-bit with the
oracle; the cases also pin WHICH route ran: the single pass where it applies, the exact two-pass route where the
kernel must decline (bytes outside the alphabet, lmax below the read length, parse errors, long reads)."""
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ALPH = np.frombuffer(b"ACGTN", dtype=np.uint8)


@pytest.fixture(scope="module")
def env():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as g
    pkg = g.load_package()
    yield torch, pkg


def make(rng, nrec, seqlen, crlf=0.0, plus_id=False, hdr=None, qlo=33, qhi=75):
    out = []
    for i in range(nrec):
        n = seqlen(i) if callable(seqlen) else seqlen
        e = b"\r\n" if rng.random() < crlf else b"\n"
        h = hdr(i) if hdr else b"r%d 1:N:0" % i
        seq = rng.choice(ALPH, n, p=[.2475, .2475, .2475, .2475, .01]).tobytes()
        qual = rng.integers(qlo, qhi, n).astype(np.uint8).tobytes()
        out.append(b"@" + h + e + seq + e + b"+" + (h if plus_id else b"") + e + qual + e)
    return b"".join(out)


def run(env, fqref, data, lmax, want_fused, offsets=False, want_route=None):
    torch, pkg = env
    # a context of its own: list sizes and the fast path's back-off stick to a context
    ctx = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    try:
        return run_ctx(torch, pkg, ctx, fqref, data, lmax, want_fused, offsets, want_route)
    finally:
        ctx.close()


def run_ctx(torch, pkg, ctx, fqref, data, lmax, want_fused, offsets, want_route=None):
    dev = torch.device("cuda:0")
    a = np.frombuffer(data, dtype=np.uint8)
    d = torch.empty(max(a.size, 16), dtype=torch.uint8, device=dev)
    d[:a.size].copy_(torch.from_numpy(a.copy()))
    qh = torch.zeros(lmax * 256, dtype=torch.int64, device=dev)
    bh = torch.zeros(lmax * 8, dtype=torch.int64, device=dev)
    sc = torch.zeros(8, dtype=torch.int64, device=dev)
    # the histograms are ADDED to: start from a known non-zero state to see that a discarded pass leaves no trace
    qh += 3
    bh += 5
    sc += 7
    ctx.set_spec(True)  # (forget any back-off an earlier case left in the context)
    ctx.set_single_pass(True)  # (... and the single pass's own, after a pass it gave up)
    rs = None
    if offsets:
        rs = torch.zeros(a.size // 4 + 16, dtype=torch.int64, device=dev)
        s, c, st = ctx.scan_stats(d.data_ptr(), a.size, lmax, qh.data_ptr(), bh.data_ptr(), sc.data_ptr(),
                                  d_rec_start=rs.data_ptr(), cap=rs.numel())
        assert st == pkg.OK
    else:
        s, c = ctx.stats(d.data_ptr(), a.size, lmax, qh.data_ptr(), bh.data_ptr(), sc.data_ptr())
    fused = ctx.last_scan_fast()
    r, oq, ob, osc = fqref.stats(a, lmax)
    assert (s.parse_status, s.n_records) == (r.status, r.n_records)
    assert np.array_equal(sc.cpu().numpy().astype(np.uint64) - 7, osc), (sc.cpu().numpy() - 7, osc)
    assert np.array_equal(qh.cpu().numpy().astype(np.uint64).reshape(lmax, 256) - 3, oq)
    assert np.array_equal(bh.cpu().numpy().astype(np.uint64).reshape(lmax, 8) - 5, ob)
    if offsets:
        r2, off = fqref.offsets(a)
        assert np.array_equal(rs.cpu().numpy()[: r2.n_records].astype(np.uint64), off)
    if want_fused is not None:
        assert fused == want_fused, "route: single pass kept = %s, expected %s" % (fused, want_fused)
        if want_route is None:
            want_route = 1 if want_fused else 0
    if want_route is not None and want_route >= 0:
        assert ctx.last_stats_route() == want_route, "statistics route %d, expected %d" % (ctx.last_stats_route(), want_route)
    return s


@pytest.mark.parametrize("nrec", [1, 7, 49, 50, 55, 199, 200, 204, 397, 398, 1000, 3177])
def test_sizes_around_tiles_and_spans(env, fqref, nrec):
    """330-byte records: buffers that end inside the first tile, at / around tile (16 KiB) and span (64 KiB) edges."""
    rng = np.random.default_rng(nrec)
    data = make(rng, nrec, 150, hdr=lambda i: b"SYN.%012d 1:N:0:1" % i)
    assert len(data) == 330 * nrec
    # (nrec 50, 199, 398: a last tile of a few hundred bytes — fewer than eight line starts — is k_finalize_fast's)
    run(env, fqref, data, 150, want_fused=nrec >= 3)


@pytest.mark.parametrize("shape", ["fixed150", "fixed100", "fixed36", "fixed250", "ragged", "ragged4", "crlf", "plusid",
                                   "qual_at_plus", "empty_reads", "long_headers36", "long_headers50", "long_plusid"])
def test_single_pass_shapes(env, fqref, shape):
    rng = np.random.default_rng(zlib.crc32(shape.encode()))
    lmaxes = (150,)
    kw = {}
    if shape == "fixed150":
        seqlen, nrec, lmaxes = 150, 5000, (150, 151, 160, 256)
    elif shape == "fixed100":
        seqlen, nrec, lmaxes = 100, 6000, (100, 128)
    elif shape == "fixed36":
        seqlen, nrec, lmaxes = 36, 20000, (36, 64)       # > 64 line starts per 4 KiB group: several chunks of entries
    elif shape == "fixed250":
        seqlen, nrec, lmaxes = 250, 4000, (250, 256)     # the 256-row variant of the kernel
    elif shape == "ragged":
        seqlen, nrec, lmaxes = (lambda i: int(rng.integers(1, 151))), 8000, (150, 160)   # batches of lines of different lengths
    elif shape == "ragged4":
        seqlen, nrec, lmaxes = (lambda i: int(rng.choice([148, 149, 150, 151, 152, 4, 3, 1]))), 8000, (152,)
    elif shape == "crlf":
        seqlen, nrec, kw = 150, 5000, {"crlf": 0.3}       # trim_winline in the kernel (src/records.rs:66-73)
    elif shape == "plusid":
        seqlen, nrec, kw = 150, 5000, {"plus_id": True}
    elif shape == "long_headers36":   # header lines longer than lmax (only sequence / quality lines must fit the rows)
        seqlen, nrec, lmaxes = 36, 20000, (36, 40)
        kw = {"hdr": lambda i: b"A00123:45:HXXXXXXXX:1:%04d:%05d:%05d 1:N:0:ATCACG" % (1101 + i % 400, 1000 + 7 * i % 30000, 2000 + 3 * i % 30000)}
    elif shape == "long_headers50":
        seqlen, nrec, lmaxes = 50, 12000, (50,)
        kw = {"hdr": lambda i: b"x" * 100 + b"%d" % i}
    elif shape == "long_plusid":      # ... and separator lines that repeat a long id
        seqlen, nrec, lmaxes = 50, 12000, (50, 64)
        kw = {"hdr": lambda i: b"SRR000001.%d length=50 some text" % i, "plus_id": True}
    elif shape == "qual_at_plus":
        seqlen, nrec, kw = 150, 5000, {"qlo": 43, "qhi": 65}   # quality lines full of '+' and '@'
    else:
        seqlen, nrec = (lambda i: 0 if i % 5 == 0 else 150), 6000  # empty sequence / quality lines are accepted
    data = make(rng, nrec, seqlen, **kw)
    for lmax in lmaxes:
        run(env, fqref, data, lmax, want_fused=True)
    run(env, fqref, data, lmaxes[0], want_fused=True, offsets=True)


@pytest.mark.parametrize("shape", ["fixed300", "fixed500", "fixed511", "ragged511", "crlf300", "mixed", "plusid300", "empty_and_long"])
def test_single_pass_counts_up_to_511_columns(env, fqref, shape):
    """Reads of 257 .. 511 columns (MiSeq 2 x 300, merged pairs) take the scan's own pass too (the reference
    treats every record up to BUFSIZE alike, src/records.rs:75-90, src/lib.rs:276-283): the packed instance of k_scan_stats —
    sixteen steps per line, two rows per LDS word, flushed before a 16-bit half can wrap.  Values == oracle, route pinned."""
    rng = np.random.default_rng(zlib.crc32(shape.encode()))
    kw = {}
    if shape == "fixed300":
        seqlen, nrec, lmaxes = 300, 3000, (300, 301, 320, 384, 512)
    elif shape == "fixed500":
        seqlen, nrec, lmaxes = 500, 2500, (500, 512)
    elif shape == "fixed511":
        seqlen, nrec, lmaxes = 511, 2500, (511, 512)
    elif shape == "ragged511":
        seqlen, nrec, lmaxes = (lambda i: int(rng.integers(0, 512))), 5000, (511, 512)
    elif shape == "crlf300":
        seqlen, nrec, lmaxes, kw = 300, 3000, (300, 400), {"crlf": 0.3}
    elif shape == "mixed":      # batches of eight hold lines on both sides of every step boundary
        seqlen, nrec, lmaxes = (lambda i: int(rng.choice([100, 150, 255, 256, 257, 259, 288, 300, 320, 383, 384, 385, 510, 511, 4, 1]))), 6000, (511,)
    elif shape == "plusid300":
        seqlen, nrec, lmaxes, kw = 300, 3000, (300,), {"plus_id": True, "hdr": lambda i: b"M01234:56:000000000-ABCDE:1:1101:%05d:%05d 1:N:0:1" % (i * 7 % 30000, i * 3 % 30000)}
    else:
        seqlen, nrec, lmaxes = (lambda i: 0 if i % 5 == 0 else 300), 4000, (300,)
    data = make(rng, nrec, seqlen, **kw)
    for lmax in lmaxes:
        run(env, fqref, data, lmax, want_fused=True)
    run(env, fqref, data, lmaxes[0], want_fused=True, offsets=True)


@pytest.mark.parametrize("nrec", [1, 2, 3, 7, 24, 25, 26, 99, 100, 101])
def test_tiny_and_tile_edge_inputs_of_300_columns(env, fqref, nrec):
    """Files of a few records of 300 columns, and files that end around tile (16 KiB) and span (64 KiB) edges, through the wide
    instances' rows (300, 320, 511): whatever route the call takes — the fast path cannot prove a file of fewer than eight
    line starts — the values are the oracle's."""
    rng = np.random.default_rng(1000 + nrec)
    data = make(rng, nrec, 300, hdr=lambda i: b"M%05d:%012d" % (i, i))   # 621-byte records: 26.4 per tile
    for lmax in (300, 320, 511):
        run(env, fqref, data, lmax, want_fused=None)
    run(env, fqref, data, 300, want_fused=None, offsets=True)


@pytest.mark.parametrize("shape", ["dirty_both_halves", "few_longer_than_lmax", "lines_of_512_and_more"])
def test_declined_counts_beyond_256_columns(env, fqref, shape):
    """... and what that pass does not count itself goes the same ways as below the 256th column: batches with a byte outside the
    alphabets are dumped with all sixteen steps (k_stats_declined counts them in two windows of 256 rows), lines beyond lmax are
    listed; a line of 512 columns or more is listed if the kept tail still holds its beginning and sends the histograms to a
    second pass if not.  The scan's result stands either way."""
    torch, pkg = env
    rng = np.random.default_rng(zlib.crc32(shape.encode()))
    lmax, route = 300, 2
    if shape == "dirty_both_halves":
        recs = [bytearray(make(rng, 1, 300, hdr=lambda i: b"r%d" % j)) for j in range(3000)]
        for j in rng.choice(3000, 300, replace=False):
            r = recs[j]
            h = r.index(b"\n") + 1
            col = int(rng.choice([0, 1, 3, 100, 255, 256, 257, 258, 259, 287, 288, 296, 299]))
            if rng.random() < 0.5:
                r[h + col] = int(rng.choice(np.frombuffer(b"acgtnRYKM.-*", dtype=np.uint8)))
            else:
                r[h + 301 + 2 + col] = int(rng.choice([97, 105, 126, 125, 200, 255, 0, 32]))
        data = b"".join(bytes(r) for r in recs)
    elif shape == "few_longer_than_lmax":
        # (none in the four 64 KiB windows the context looks at — the input's first bytes and three more, a quarter of it apart:
        # the pass keeps rows for 300 columns and LISTS the longer lines)
        data = make(rng, 3000, lambda i: 420 if i % 400 == 199 else 300)
    else:
        data = make(rng, 3000, lambda i: int(rng.choice([512, 513, 600])) if i % 300 == 11 else 300)
        lmax, route = 512, -1        # (listed or declined, by where the line happens to lie in its 4 KiB group)
    ctx = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    try:
        for _ in range(2):
            run_ctx(torch, pkg, ctx, fqref, data, lmax, True, False, want_route=route)
        if route < 0:
            assert ctx.last_stats_route() in (0, 2)
        a = np.frombuffer(data, dtype=np.uint8)
        d = torch.from_numpy(a.copy()).cuda()
        s = ctx.scan(d.data_ptr(), a.size)[0]
        assert ctx.last_scan_fast() and s.parse_status == pkg.OK
    finally:
        ctx.close()


@pytest.mark.parametrize("shape", ["lowercase", "qual_high", "both_many", "few_long", "crlf_dirty", "lmax_short", "long_reads"])
def test_declined_counts_are_no_parse_doubt(env, fqref, shape):
    """Any byte may stand in seq() / qual() (src/records.rs:19-33, 75-90).  What the single pass does not count itself — a
    batch of eight lines with a byte outside ACGTN / '!'..'`', a line longer than lmax — is counted one by one behind it
    (route 2); lines of more than ~500 bytes, or more declined lines than the dump area holds, send the HISTOGRAMS to a second
    pass (route 0).  Either way the scan's result stands on the fast path, the arrays get exactly the oracle's counts, and
    the context's next plain scan takes the fast path: nothing was wrong with the parse."""
    torch, pkg = env
    rng = np.random.default_rng(11)
    data = bytearray(make(rng, 4000, 150))
    lmax, route = 150, 2
    if shape == "lowercase":
        pos = data.index(b"\n", 700000) + 1   # some line start far into the file; flip a base of the next sequence line
        k = data.index(b"\n", pos) + 5
        data[k] = ord("a")
    elif shape == "qual_high":
        k = len(data) - 20
        data[k] = 126                         # '~' is outside the kernel's window '!'..'`' (and a valid quality byte)
    elif shape == "both_many":                # one byte in ~3000 of either kind, anywhere in the lines (first, last, partial dwords)
        recs = [bytearray(make(rng, 1, 150, hdr=lambda i: b"r%d" % j)) for j in range(4000)]
        for j in rng.choice(4000, 400, replace=False):
            r = recs[j]
            h = r.index(b"\n") + 1
            col = int(rng.integers(0, 150))
            if rng.random() < 0.5:
                r[h + col] = int(rng.choice(np.frombuffer(b"acgtnRYKM.-*", dtype=np.uint8)))
            else:
                r[h + 151 + 2 + col] = int(rng.choice([97, 105, 126, 125, 200, 255, 0, 32]))   # above '`', below '!', NUL, high bit
        data = bytearray(b"".join(bytes(r) for r in recs))
    elif shape == "few_long":                 # a few reads longer than the pass's rows among reads that fit: listed, columns >= lmax go to scalars[5], [6]
        data = bytearray(make(rng, 4000, lambda i: 190 if i % 500 == 299 else 150))   # (none in the four 64 KiB windows by which the pass sizes its rows)
    elif shape == "crlf_dirty":
        data = bytearray(make(rng, 4000, 150, crlf=0.5))
        for k in rng.integers(1000, len(data) - 1000, 40):
            if data[k] not in (10, 13, 43, 64):   # (not a line end, and no new '+' / '@' at a line start)
                data[k] = ord("n") if data[k] in b"ACGTN" else data[k] | 0x80
    elif shape == "lmax_short":
        lmax, route = 100, 1                  # EVERY line is longer than lmax: the pass keeps rows for the reads, columns 100 .. 149 become scalars[5], [6]
    elif shape == "long_reads":
        data = bytearray(make(rng, 600, 700))  # lines longer than the tail the kernel keeps
        lmax, route = 256, 0
    ctx = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    try:
        for _ in range(2):   # (the second call: no back-off from the first)
            r = fqref.count(bytes(data))
            assert r.status == pkg.OK
            # (reads of 700 bases: a look at the input's first 64 KiB tells the context that most lines are longer than the pass
            # takes, and the call is counted over the exact index without an attempt — the scan then is the exact path's)
            run_ctx(torch, pkg, ctx, fqref, bytes(data), lmax, None if shape == "long_reads" else True, False, want_route=route)
        a = np.frombuffer(bytes(data), dtype=np.uint8)
        d = torch.from_numpy(a.copy()).cuda()
        s = ctx.scan(d.data_ptr(), a.size)[0]
        assert ctx.last_scan_fast() and (s.parse_status, s.n_records) == (pkg.OK, r.n_records)
    finally:
        ctx.close()
    if shape in ("lowercase", "both_many"):
        run(env, fqref, bytes(data), lmax, want_fused=True, offsets=True, want_route=route)


@pytest.mark.parametrize("shape", ["mismatch", "truncated", "no_final_nl", "mismatch_and_dirty"])
def test_parse_errors_take_the_exact_route(env, fqref, shape):
    """What the single pass cannot PROVE sends the call to the exact two-pass route; the arrays the caller passed get exactly
    the oracle's counts (nothing of the discarded pass — not even the lines it had dumped for a recount — leaks into them)."""
    rng = np.random.default_rng(11)
    data = bytearray(make(rng, 4000, 150))
    if shape in ("mismatch", "mismatch_and_dirty"):
        k = data.index(b"\n", 900000)
        del data[k - 3]                       # one quality (or sequence) byte less: length mismatch somewhere
        if shape == "mismatch_and_dirty":
            data[data.index(b"\n", data.index(b"\n", 300000) + 1) - 9] = ord("c")
    elif shape == "truncated":
        del data[-100:]
    elif shape == "no_final_nl":
        del data[-1:]
    run(env, fqref, bytes(data), 150, want_fused=False)


@pytest.mark.parametrize("crlf", [0.0, 0.3])
def test_short_last_tile(env, fqref, crlf):
    """Files that end a few bytes to a few hundred bytes into a 16 KiB tile: the last tile has 0..7 line starts, which
    no window of five can vouch for.  The per-tile kernels mark it (FR_SMALL) and k_finalize_fast validates and emits
    its records; the single pass counts its lines with the span in front of it.  Both routes must keep the fast
    path, and give the oracle's offsets and histograms."""
    torch, pkg = env
    rng = np.random.default_rng(5 + int(crlf * 10))
    recs = [make(rng, 1, 150 - (k % 3), crlf=crlf, hdr=lambda i: b"q%d" % (k * 7919 % 1000)) for k in range(700)]
    sizes = np.cumsum([len(r) for r in recs])
    picked = [k for k in range(120, 700) if sizes[k - 1] % 16384 < 1700]
    assert len(picked) >= 30
    for k in picked:
        data = b"".join(recs[:k])
        run(env, fqref, data, 150, want_fused=True, offsets=bool(k & 1))
        # the plain scan on its own takes the same route
        ctx = pkg.Ctx(0)
        a = np.frombuffer(data, dtype=np.uint8)
        d = torch.from_numpy(a.copy()).cuda()
        rs = torch.zeros(k + 16, dtype=torch.int64, device=d.device)
        s, c = ctx.scan(d.data_ptr(), a.size, d_rec_start=rs.data_ptr(), cap=rs.numel())[:2]
        r2, off = fqref.offsets(a)
        assert ctx.last_scan_fast() and (s.parse_status, s.n_records) == (r2.status, r2.n_records) == (pkg.OK, k)
        assert np.array_equal(rs.cpu().numpy()[: k + 1].astype(np.uint64), np.append(off, a.size).astype(np.uint64)[: k + 1])
        ctx.close()
    # ... and tails of exactly t bytes, for the cuts between and inside the last record's lines
    for t in (1, 2, 3, 150, 151, 152, 153, 154, 155, 156, 157, 158, 303, 304, 305, 306, 307, 308, 309, 310, 311, 312, 313,
              600, 620, 640):
        T0 = 120 * (4 + 151 + 2 + 151)
        q, r = divmod((t - T0) % 16384, 120)
        data = make(rng, 120, 150, crlf=0.0, hdr=lambda i: b"h:" + b"p" * (q + (1 if i < r else 0)))
        assert len(data) % 16384 == t
        run(env, fqref, data, 150, want_fused=True, offsets=True)


def test_short_last_tile_with_errors(env, fqref):
    """An error inside the short last tile (or in the record that straddles into it) is the exact route's to report."""
    rng = np.random.default_rng(77)
    base = make(rng, 120, 150, hdr=lambda i: b"q%d" % i)
    while len(base) % 16384 > 900 or len(base) % 16384 < 400:
        base += make(rng, 1, 150, hdr=lambda i: b"x")
    n = len(base)
    tail0 = n - n % 16384
    for kind in ("at", "plus", "len", "trunc", "nonl"):
        data = bytearray(base)
        if kind == "at":
            k = data.rindex(b"\n@") + 1
            assert k >= tail0
            data[k] = ord("x")
        elif kind == "plus":
            k = data.rindex(b"\n+\n") + 1
            data[k] = ord("-")
        elif kind == "len":
            del data[n - 3]
        elif kind == "trunc":
            del data[n - 160:]
        else:
            del data[n - 1:]
        run(env, fqref, bytes(data), 150, want_fused=False)


def test_line_list_workspace_is_allocated_on_demand(env, fqref):
    """The fast path writes one or two 128-byte lines per 16 KiB tile; the 1 KiB-per-tile line lists are the exact path's
    (and the fast path's overflow area for reads shorter than ~50 bp).  A context that only sees ordinary reads on the
    fast path never allocates them: 6 % of the input size stays free."""
    torch, pkg = env
    dev = torch.device("cuda:0")
    n = (2 << 30) // 330 * 330
    d = torch.empty(n + 16, dtype=torch.uint8, device=dev)
    ctx = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    ctx.synth_fill(d.data_ptr(), 0, n)
    rs = torch.zeros(n // 300 + 16, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    s, c, st = ctx.scan(d.data_ptr(), n, True, None, rs.data_ptr(), rs.numel())
    assert s.n_records == n // 330 and ctx.last_scan_fast()
    used = free0 - torch.cuda.mem_get_info()[0]
    assert used < (100 << 20), used          # two lines per tile = 32 MiB + change; the line lists alone would be 128 MiB
    # 36 bp reads: ~150 records per tile, more than a tile's two lines hold: the context gets its line lists, keeps the fast path
    rng = np.random.default_rng(3)
    short = make(rng, 60000, 36)
    a = np.frombuffer(short, dtype=np.uint8)
    d2 = torch.from_numpy(a.copy()).to(dev)
    s2, c2, st2 = ctx.scan(d2.data_ptr(), a.size, True, None, rs.data_ptr(), rs.numel())
    r2, off = fqref.offsets(a)
    assert (s2.parse_status, s2.n_records) == (r2.status, r2.n_records) == (pkg.OK, 60000) and ctx.last_scan_fast()
    assert np.array_equal(rs.cpu().numpy()[:60000].astype(np.uint64), off)
    s, c, st = ctx.scan(d.data_ptr(), n, True, None, rs.data_ptr(), rs.numel())   # (now with the lists: 128 MiB more)
    assert s.n_records == n // 330 and ctx.last_scan_fast()
    assert free0 - torch.cuda.mem_get_info()[0] > (128 << 20)
    ctx.close()


def test_medium_synthetic_parity_and_determinism(env, fqref):
    """256 MiB of the benchmark's synthetic input: single pass == oracle, and two runs give identical arrays."""
    torch, pkg = env
    ctx = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    dev = torch.device("cuda:0")
    n = (256 << 20) // 330 * 330
    d = torch.empty(n + 16, dtype=torch.uint8, device=dev)
    ctx.synth_fill(d.data_ptr(), 0, n)
    host = d[:n].cpu().numpy()
    res = []
    for _ in range(2):
        qh = torch.zeros(150 * 256, dtype=torch.int64, device=dev)
        bh = torch.zeros(150 * 8, dtype=torch.int64, device=dev)
        sc = torch.zeros(8, dtype=torch.int64, device=dev)
        rs = torch.zeros(n // 300 + 16, dtype=torch.int64, device=dev)
        ctx.set_spec(True)
        s, c, st = ctx.scan_stats(d.data_ptr(), n, 150, qh.data_ptr(), bh.data_ptr(), sc.data_ptr(),
                                  d_rec_start=rs.data_ptr(), cap=rs.numel())
        assert ctx.last_scan_fast() and s.parse_status == pkg.OK and s.n_records == n // 330
        res.append((qh.cpu().numpy(), bh.cpu().numpy(), sc.cpu().numpy(), rs.cpu().numpy()[: n // 330 + 1]))
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)
    r, oq, ob, osc = fqref.stats(host, 150)
    assert np.array_equal(res[0][0].astype(np.uint64).reshape(150, 256), oq)
    assert np.array_equal(res[0][1].astype(np.uint64).reshape(150, 8), ob)
    assert np.array_equal(res[0][2].astype(np.uint64), osc)
    assert np.array_equal(res[0][3].astype(np.uint64), np.arange(n // 330 + 1, dtype=np.uint64) * 330)
    ctx.close()
