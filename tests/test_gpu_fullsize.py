"""BASELINE.json full size (configs[1]/[2]: 16 GiB of synthetic 150 bp FASTQ resident in HBM),
checked through size-independent properties: analytic record count and offsets, per-position
histogram totals, linearity (histogram of the whole == sum of histograms of record-aligned parts),
an oracle cross-check on a sub-range, and error injection deep inside the buffer."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RECLEN = 330
NBYTES = (1 << 34) // RECLEN * RECLEN  # 17 179 868 970
NREC = NBYTES // RECLEN                # 52 060 209


@pytest.fixture(scope="module")
def env():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as g
    pkg = g.load_package()
    dev = torch.device("cuda:0")
    ctx = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    buf = torch.empty(NBYTES + 16, dtype=torch.uint8, device=dev)
    ctx.synth_fill(buf.data_ptr(), 0, NBYTES)
    yield torch, pkg, ctx, buf, dev
    ctx.close()


def hists(torch, dev):
    return (torch.zeros(150 * 256, dtype=torch.int64, device=dev),
            torch.zeros(150 * 8, dtype=torch.int64, device=dev),
            torch.zeros(8, dtype=torch.int64, device=dev))


def test_count_and_offsets_16gib(env):
    torch, pkg, ctx, buf, dev = env
    rs = torch.empty(NREC + 1, dtype=torch.int64, device=dev)
    s, c, st = ctx.scan(buf.data_ptr(), NBYTES, True, None, rs.data_ptr(), NREC + 1)
    assert st == pkg.OK and s.parse_status == pkg.OK
    assert s.n_records == NREC == 52060209
    assert s.n_newlines == 4 * NREC and s.tail_len == 0 and s.max_record_len == RECLEN
    assert s.bytes_consumed == NBYTES
    assert torch.equal(rs, torch.arange(NREC + 1, dtype=torch.int64, device=dev) * RECLEN)
    s2, c2, st = ctx.scan(buf.data_ptr(), NBYTES, True, None, None, 0)  # count-only
    assert (s2.n_records, s2.parse_status, s2.bytes_consumed) == (NREC, pkg.OK, NBYTES)
    assert (c.base_offset, c.nl_count, list(c.back)) == (NBYTES, 4 * NREC, [0, 151, 153, 304])


def test_truncated_and_injected_errors_16gib(env, fqref):
    torch, pkg, ctx, buf, dev = env
    s, c, st = ctx.scan(buf.data_ptr(), NBYTES - 100, True, None, None, 0)
    assert (s.parse_status, s.n_records, s.err_record) == (pkg.E_TRUNCATED, NREC - 1, NREC - 1)
    assert s.bytes_consumed == NBYTES - RECLEN and s.tail_len == RECLEN - 100
    k = 40_000_123
    for off, expect in ((0, pkg.E_HEADER), (177, pkg.E_SEP), (176, pkg.E_LEN_MISMATCH)):
        pos = k * RECLEN + off
        old = int(buf[pos].item())
        buf[pos] = ord("x")
        s, c, st = ctx.scan(buf.data_ptr(), NBYTES, True, None, None, 0)
        buf[pos] = old
        # the exact status, also for the swallowed newline (off 176: the sequence line swallows "+", the '+' check of the merged
        # record fails first, src/records.rs:155): the oracle over the records k .. k + 2 with the same byte changed — the parser's
        # verdict on record k depends on nothing in front of it
        win = bytearray(fqref.synth(k * RECLEN, 3 * RECLEN).tobytes())
        win[off] = ord("x")
        want = fqref.count(bytes(win))
        assert (want.status, want.n_records) == (expect if off != 176 else want.status, 0) and want.status != 0
        assert s.parse_status == want.status
        assert (s.n_records, s.err_record, s.err_offset) == (k, k, k * RECLEN)
    s, c, st = ctx.scan(buf.data_ptr(), NBYTES, True, None, None, 0)
    assert s.parse_status == pkg.OK and s.n_records == NREC


def test_histograms_16gib_properties(env, fqref):
    torch, pkg, ctx, buf, dev = env
    qh, bh, sc = hists(torch, dev)
    s, c = ctx.stats(buf.data_ptr(), NBYTES, 150, qh.data_ptr(), bh.data_ptr(), sc.data_ptr())
    assert s.n_records == NREC
    q = qh.view(150, 256)
    b = bh.view(150, 8)
    assert torch.all(q.sum(dim=1) == NREC) and torch.all(b.sum(dim=1) == NREC)
    assert int(q[:, :35].sum()) == 0 and int(q[:, 74:].sum()) == 0  # '#'..'I' only
    assert int(b[:, 5:].sum()) == 0
    scal = sc.cpu().numpy()
    assert list(scal[:3]) == [NREC, NREC * 150, NREC * 150] and scal[5] == scal[6] == 0
    assert scal[4] == NREC and 0 < scal[3] < NREC  # all ACGTN; some reads contain N
    # linearity: whole == sum of 4 record-aligned (and 16-byte aligned) parts
    step = (NREC // 4) // 16 * 16
    cuts = [0, step, 2 * step, 3 * step, NREC]
    q2, b2, s2 = hists(torch, dev)
    for a, e in zip(cuts[:-1], cuts[1:]):
        ctx.stats(buf.data_ptr() + a * RECLEN, (e - a) * RECLEN, 150, q2.data_ptr(), b2.data_ptr(), s2.data_ptr())
    assert torch.equal(q2, qh) and torch.equal(b2, bh) and torch.equal(s2, sc)
    # oracle cross-check on a 32 MiB record-aligned sub-range deep inside the buffer
    r0, nr = 30_000_000 // 16 * 16, (32 << 20) // RECLEN
    q3, b3, s3 = hists(torch, dev)
    ctx.stats(buf.data_ptr() + r0 * RECLEN, nr * RECLEN, 150, q3.data_ptr(), b3.data_ptr(), s3.data_ptr())
    host = buf[r0 * RECLEN: (r0 + nr) * RECLEN].cpu().numpy()
    assert np.array_equal(host, fqref.synth(r0 * RECLEN, nr * RECLEN))
    r, oq, ob, osc = fqref.stats(host, 150)
    assert np.array_equal(q3.cpu().numpy().astype(np.uint64).reshape(150, 256), oq)
    assert np.array_equal(b3.cpu().numpy().astype(np.uint64).reshape(150, 8), ob)
    assert np.array_equal(s3.cpu().numpy().astype(np.uint64), osc)


def test_rows_on_either_side_of_the_reads_16gib(env):
    """lmax is the caller's choice (the closure of Parser::each has none, src/lib.rs:226-237): 1000 rows over the 150-base reads,
    and 100.  One pass either way (the pass keeps the rows the READS need); the first 100 / 150 rows equal those of the call
    with 150 rows, rows beyond the reads stay zero, and the columns beyond 100 rows are the overflow counters."""
    torch, pkg, ctx, buf, dev = env
    q0, b0, s0 = hists(torch, dev)
    ctx.invalidate()
    ctx.stats(buf.data_ptr(), NBYTES, 150, q0.data_ptr(), b0.data_ptr(), s0.data_ptr())
    assert ctx.last_stats_route() == 1 and int(s0[0]) == NREC
    for lmax in (1000, 100):
        q = torch.zeros(lmax * 256, dtype=torch.int64, device=dev)
        b = torch.zeros(lmax * 8, dtype=torch.int64, device=dev)
        sc = torch.zeros(8, dtype=torch.int64, device=dev)
        ctx.invalidate()
        s, c = ctx.stats(buf.data_ptr(), NBYTES, lmax, q.data_ptr(), b.data_ptr(), sc.data_ptr())
        assert (s.parse_status, s.n_records) == (pkg.OK, NREC) and ctx.last_stats_route() == 1, (lmax, ctx.last_stats_route())
        k = min(lmax, 150)
        assert torch.equal(q.view(lmax, 256)[:k], q0.view(150, 256)[:k]) and torch.equal(b.view(lmax, 8)[:k], b0.view(150, 8)[:k])
        assert int(q.view(lmax, 256)[k:].sum()) == 0 and int(b.view(lmax, 8)[k:].sum()) == 0
        want = s0.clone()
        want[5] = want[6] = NREC * (150 - k)
        assert torch.equal(sc, want), (lmax, sc.cpu().tolist(), want.cpu().tolist())


def test_packed_rows_do_not_wrap_12gib(env):
    """Reads of 300 columns at full size through the packed instance of the single pass (two rows per LDS word, 16-bit halves,
    flushed every few spans): 12 GiB of ONE record repeated — every quality byte 'I', every base 'A' — puts ~ 76 000 counts per
    block on a single (column, bin), more than a 16-bit half holds: a flush that came late would wrap.  Properties that need no
    oracle: every row's one bin holds the record count exactly (src/records.rs:75-90 — the consumer's loop over seq() / qual())."""
    torch, pkg, ctx, buf, dev = env
    L = 300
    rec = b"@M01234:56:000000000-ABCDE:1:1101:12345:6789 1:N:0:1\n" + b"A" * L + b"\n+\n" + b"I" * L + b"\n"
    reps_blk = 4096
    blk = torch.from_numpy(np.frombuffer(rec * reps_blk, dtype=np.uint8).copy()).to(dev)
    n_blk = (12 << 30) // blk.numel()
    n = n_blk * blk.numel()
    assert n <= NBYTES
    view = buf[:n].view(n_blk, blk.numel())
    view.copy_(blk.unsqueeze(0).expand(n_blk, -1))          # (the module's buffer, overwritten: this test runs last in the file)
    nrec = n_blk * reps_blk
    qh = torch.zeros(L * 256, dtype=torch.int64, device=dev)
    bh = torch.zeros(L * 8, dtype=torch.int64, device=dev)
    sc = torch.zeros(8, dtype=torch.int64, device=dev)
    ctx.invalidate()
    s, c = ctx.stats(buf.data_ptr(), n, L, qh.data_ptr(), bh.data_ptr(), sc.data_ptr())
    assert (s.parse_status, s.n_records) == (pkg.OK, nrec) and ctx.last_stats_route() == 1 and ctx.last_scan_fast()
    assert nrec // 256 > 65535                              # (a CU's share of the records does not fit a 16-bit half)
    q, b = qh.view(L, 256), bh.view(L, 8)
    assert torch.all(q[:, ord("I")] == nrec) and int(q.sum()) == nrec * L
    assert torch.all(b[:, 0] == nrec) and int(b.sum()) == nrec * L
    assert sc.cpu().tolist()[:5] == [nrec, nrec * L, nrec * L, nrec, nrec]
    ctx.synth_fill(buf.data_ptr(), 0, NBYTES)               # (leave the buffer as the fixture made it)
    ctx.invalidate()


@pytest.mark.parametrize("varied", [False, True])
def test_kilobase_reads_12gib(varied):
    """configs[2]'s per-read statistics loop over KILOBASE reads at full size (12 GiB; the 16 GiB buffer of this module is
    somebody else's): reads of 5 kbp, and reads of log-normal length around 5 kbp (370 .. 30 000 bases: 118 column blocks,
    k_stats_long's work items cut on the device from a census of the lengths, per-column-block record lists), against the
    per-position count numpy makes of the repeated block — every (position, quality value) counter and the base total,
    bit-exact (src/records.rs:75-90; src/lib.rs:276-283 treats every record up to the Buffer alike).  bench.py's own leg, at
    three times its size."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as g
    import bench
    pkg = g.load_package()
    dev = torch.device("cuda:0")
    ctx = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    try:
        r = bench.long_read_leg(pkg, torch, dev, ctx, gib=12.0, varied=varied)   # (asserts the counts itself)
        assert r["gbs_end_to_end"] > 0
    finally:
        ctx.close()


@pytest.mark.parametrize("config", ["configs[1] scan", "configs[2] stats"])
def test_a_fresh_context_takes_the_expected_route_16gib(env, config):
    """The routes are heuristics with history (fast-path back-off, single-pass hints, adaptive line buffers); results never
    depend on them, speed does.  A FRESH context on each BASELINE config must take the fast route at its FIRST call — the
    fast path's byte scan for configs[1], the single pass (k_scan_stats) for configs[2] — so that a heuristic that regresses
    shows up here as a failure, not in the bench as a slower number."""
    torch, pkg, ctx, buf, dev = env
    fresh = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    try:
        if config.startswith("configs[1]"):
            rs = torch.empty(NREC + 1, dtype=torch.int64, device=dev)
            s, c, st = fresh.scan(buf.data_ptr(), NBYTES, True, None, rs.data_ptr(), NREC + 1)
            assert (st, s.parse_status, s.n_records) == (pkg.OK, pkg.OK, NREC)
            assert fresh.last_scan_fast() == 1
            lb = fresh.line_buffers()
            assert lb["alive"] == 1, lb   # one line buffer after one call: nothing was searched or tried
        else:
            qh, bh, sc = hists(torch, dev)
            s, c = fresh.stats(buf.data_ptr(), NBYTES, 150, qh.data_ptr(), bh.data_ptr(), sc.data_ptr())
            assert (s.parse_status, s.n_records) == (pkg.OK, NREC)
            assert fresh.last_stats_route() == 1 and fresh.last_scan_fast() == 1
            assert int(sc[0].item()) == NREC
            # ... and so does the same call with rows far above the reads (one tool, 1000 rows, whatever comes)
            q2 = torch.zeros(1000 * 256, dtype=torch.int64, device=dev)
            b2 = torch.zeros(1000 * 8, dtype=torch.int64, device=dev)
            s2 = torch.zeros(8, dtype=torch.int64, device=dev)
            fresh2 = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
            fresh2.stats(buf.data_ptr(), NBYTES, 1000, q2.data_ptr(), b2.data_ptr(), s2.data_ptr())
            assert fresh2.last_stats_route() == 1
            fresh2.close()
    finally:
        fresh.close()
