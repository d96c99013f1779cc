"""GPU parity tests proper (-m gpu): the HIP path, called through the C ABI (include/fastq_hip.h),
against the CPU oracle on the same bytes — bit-exact record counts, offsets, index entries, error
kind / error record, and integer histograms.  Mirrors the reference's own tests
(src/lib.rs:611-811) plus the edge cases they do not pin."""
import json
import os

import numpy as np
import pytest

import fuzzgen

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_unit_tests.json")) as f:
    GOLD = json.load(f)


def build_input(parts):
    out = bytearray()
    for p in parts:
        out += p[1].encode("latin-1") * (p[2] if p[0] == "rep" else 1)
    return bytes(out)


@pytest.fixture(scope="module")
def torch():
    import torch as t
    if not t.cuda.is_available():
        pytest.skip("no GPU")
    return t


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


@pytest.fixture(scope="module")
def ctx(torch, pkg):
    c = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    yield c
    c.close()


class Gpu:
    """Small helper: host bytes -> device tensor -> fqh_scan / fqh_stats -> numpy."""

    def __init__(self, torch, ctx, pkg):
        self.t, self.ctx, self.pkg = torch, ctx, pkg
        self.dev = torch.device("cuda:0")

    def upload(self, data):
        a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        n = a.size
        d = self.t.empty(max(n, 16), dtype=self.t.uint8, device=self.dev)
        if n:
            d[:n].copy_(self.t.from_numpy(a.copy()))
        return d, n

    def scan(self, data, is_final=True, carry=None, bufsize=None, want_offsets=True):
        d, n = self.upload(data)
        return self.scan_dev(d, n, is_final, carry, bufsize, want_offsets)

    def scan_dev(self, d, n, is_final=True, carry=None, bufsize=None, want_offsets=True, off=0):
        self.ctx.set_bufsize(self.pkg.BUFSIZE if bufsize is None else bufsize)
        cap = n // 6 + 3
        rs = self.t.zeros(cap, dtype=self.t.int64, device=self.dev) if want_offsets else None
        s, c, st = self.ctx.scan(d.data_ptr() + off, n, is_final, carry,
                                 rs.data_ptr() if want_offsets else None, cap if want_offsets else 0)
        assert st == self.pkg.OK
        offs = rs.cpu().numpy().astype(np.uint64)[: s.n_records + 1] if want_offsets else None
        return s, c, offs

    def stats(self, data, lmax, bufsize=None):
        d, n = self.upload(data)
        self.ctx.set_bufsize(self.pkg.BUFSIZE if bufsize is None else bufsize)
        qh = self.t.zeros(lmax * 256, dtype=self.t.int64, device=self.dev)
        bh = self.t.zeros(lmax * 8, dtype=self.t.int64, device=self.dev)
        sc = self.t.zeros(8, dtype=self.t.int64, device=self.dev)
        s, c = self.ctx.stats(d.data_ptr(), n, lmax, qh.data_ptr(), bh.data_ptr(), sc.data_ptr())
        return (s, qh.cpu().numpy().astype(np.uint64).reshape(lmax, 256),
                bh.cpu().numpy().astype(np.uint64).reshape(lmax, 8), sc.cpu().numpy().astype(np.uint64))


@pytest.fixture(scope="module")
def gpu(torch, ctx, pkg):
    return Gpu(torch, ctx, pkg)


def assert_scan_equal(fqref, gpu, data, bufsize=None):
    B = fqref.BUFSIZE if bufsize is None else bufsize
    res, idx = fqref.index(data, bufsize=B)
    s, c, offs = gpu.scan(data, bufsize=B)
    ctxt = (B, bytes(data[:200]), len(data))
    assert s.parse_status == res.status, ctxt
    assert s.n_records == res.n_records, ctxt
    assert np.array_equal(offs[:-1], idx[:, 0]), ctxt
    if res.n_records:
        assert int(offs[-1]) == res.bytes_consumed == s.bytes_consumed, ctxt
    if res.status != fqref.OK:
        assert s.err_record == res.n_records, ctxt
    return s, c, offs, res, idx


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", [v["name"] for v in GOLD["vectors"]])
def test_reference_unit_test_vectors(fqref, gpu, name):
    """Every golden vector of the reference's own tests, through the HIP path."""
    v = next(x for x in GOLD["vectors"] if x["name"] == name)
    data = build_input(v["input"])
    s, c, offs, res, idx = assert_scan_equal(fqref, gpu, data)
    exp = v["expect"]
    assert (s.parse_status == gpu.pkg.OK) == exp["ok"]
    if "n_records" in exp:
        assert s.n_records == exp["n_records"]
    if "total_records" in exp:
        assert s.n_records == exp["total_records"]
    if "sum_of_worker_counts" in exp:
        assert s.n_records == exp["sum_of_worker_counts"]


def test_index_records_match_idxrecord(fqref, gpu, torch):
    """fqh_index_records == IdxRecord{head,seq,sep,qual,data} of src/records.rs:240-246."""
    rng = np.random.default_rng(5)
    data = fuzzgen.valid_file(rng, 500, maxlen=120)
    s, c, offs, res, idx = assert_scan_equal(fqref, gpu, data)
    out = torch.zeros(res.n_records * 3, dtype=torch.int64, device=gpu.dev)  # 24 bytes per record
    gpu.ctx.index_records(out.data_ptr(), res.n_records)
    raw = out.cpu().numpy().view(np.uint8).reshape(res.n_records, 24)
    start = raw[:, :8].copy().view(np.uint64)[:, 0]
    rest = raw[:, 8:].copy().view(np.uint32)
    assert np.array_equal(start, idx[:, 0])
    assert np.array_equal(rest.astype(np.uint64), idx[:, 1:])


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_scan_equals_oracle(fqref, gpu, seed):
    for tag, data in fuzzgen.corpus(1000 + seed, 150):
        assert_scan_equal(fqref, gpu, data)
        assert_scan_equal(fqref, gpu, data, bufsize=64)  # cfg(fuzzing) BUFSIZE, src/lib.rs:126-127


def test_edge_cases(fqref, gpu):
    cases = [b"", b"\n", b"@", b"@\n", b"@a\nAC\n+\nII\n\n", b"@a\n\n+\n\n", b"@a\nAC\r\n+\nIII\n",
             b"@a\nAC\n-\nII\n", b"@a\nAC\n", b"@a\nAC\n+", b"@a\nAC\nx", b"xa\nAC\n+\nII\n",
             b"@a\nAC\n+\n@I\n@b\nGG\n+\n+@\n", b"@\n\n+\n\n" * 7, b"\n" * 40, b"@a\nAC\n+\nII",
             b"@a\nACGT\n+\nIII\n@b\nAC\n+\nII\n"]
    for d in cases:
        assert_scan_equal(fqref, gpu, d)
        assert_scan_equal(fqref, gpu, d, bufsize=64)


def test_tile_and_piece_boundaries(fqref, gpu):
    """Records placed so that newlines, '@' and '+' fall on 16 B / 1 KiB / 16 KiB boundaries."""
    rng = np.random.default_rng(11)
    for target in (16, 1024, 16384, 32768):
        for delta in range(-3, 4):
            pre = fuzzgen.valid_file(rng, 3, crlf=False)
            pad = target + delta - len(pre) - 4  # "@" + pad*h + "\n" puts the header newline near target
            if pad < 1:
                pad += 16384
            rec = b"@" + b"h" * pad + b"\nACGT\n+\nIIII\n"
            data = pre + rec + fuzzgen.valid_file(rng, 3, crlf=False)
            assert_scan_equal(fqref, gpu, data)
            assert_scan_equal(fqref, gpu, data[: target + delta + 8])


def test_too_long_band(fqref, gpu):
    """<= B-15 always accepted, > B always rejected, alignment dependent in between."""
    rng = np.random.default_rng(3)
    B = fqref.BUFSIZE
    statuses = set()
    for trial in range(24):
        parts = [fuzzgen.valid_record(rng, j) for j in range(int(rng.integers(0, 5)))]
        L = B + int(rng.integers(-20, 3))
        parts.append(b"@" + b"h" * (L - 9) + b"\nA\n+\nB\n")
        parts += [fuzzgen.valid_record(rng, j) for j in range(int(rng.integers(0, 3)))]
        data = b"".join(parts)
        if trial % 4 == 3:
            data = data[: len(data) - int(rng.integers(1, 12))]
        s, *_ = assert_scan_equal(fqref, gpu, data)
        statuses.add(s.parse_status)
    assert gpu.pkg.E_TOO_LONG in statuses and gpu.pkg.OK in statuses


@pytest.mark.parametrize("delta", [16384, 32768, -16384])
def test_length_mismatch_by_a_multiple_of_the_tile_size(fqref, gpu, delta):
    """The tile index keeps 14-bit offsets; two raw line lengths that differ by exactly 16 384 must still differ (src/records.rs:233):
    a quality line one or two tiles longer than its sequence line, early in the file and in the middle of a tile, with tiles full
    of ordinary records on either side (no tile is short of line starts)."""
    def rec(i, sl, ql):
        return b"@r%d\n" % i + b"A" * sl + b"\n+\n" + b"I" * ql + b"\n"
    for where in (3, 700):
        parts = [rec(i, 150, 150) for i in range(1500)]
        parts[where] = rec(where, 150, 150 + delta) if delta > 0 else rec(where, 150 - delta, 150)
        s, *_ = assert_scan_equal(fqref, gpu, b"".join(parts))
        assert (s.parse_status, s.n_records) == (gpu.pkg.E_LEN_MISMATCH, where)


def test_huge_line_without_newline(fqref, gpu):
    data = b"@" + b"longid" * 200000  # 1.2 MB, no newline at all (huge_incomplete at scale)
    s, *_ = assert_scan_equal(fqref, gpu, data)
    assert s.parse_status == gpu.pkg.E_TOO_LONG


def test_dense_newlines_force_list_rerun(fqref, gpu):
    """More than 2048 line starts in one 16 KiB tile: the library reruns with full-size lists."""
    data = b"@\n\n+\n\n" * 6000
    s, *_ = assert_scan_equal(fqref, gpu, data)
    assert s.n_records == 6000


@pytest.mark.parametrize("seed", range(3))
def test_fuzz_stats_equal_oracle(fqref, gpu, seed):
    for tag, data in fuzzgen.corpus(2000 + seed, 60):
        for lmax in (8, 150):
            r, qh, bh, sc = fqref.stats(data, lmax)
            s, gq, gb, gs = gpu.stats(data, lmax)
            assert (s.parse_status, s.n_records) == (r.status, r.n_records), data
            assert np.array_equal(gq, qh) and np.array_equal(gb, bh) and np.array_equal(gs, sc), data


def test_stats_long_reads_and_odd_bytes(fqref, gpu):
    """Reads longer than the LDS-resident columns, quality bytes outside the LDS window, CRLF."""
    rng = np.random.default_rng(9)
    recs = []
    for i in range(300):
        n = int(rng.integers(0, 400))
        seq = bytes(rng.choice(list(b"ACGTNacgtn*"), n).tolist())
        qual = bytes(rng.integers(33, 127, n).astype(np.uint8).tolist()) if i % 3 else bytes([200]) * n
        e = b"\r\n" if i % 5 == 0 else b"\n"
        recs.append(b"@r" + e + seq + e + b"+" + e + qual + e)
    data = b"".join(recs)
    for lmax in (100, 300, 500):
        r, qh, bh, sc = fqref.stats(data, lmax)
        s, gq, gb, gs = gpu.stats(data, lmax)
        assert s.n_records == r.n_records == 300
        assert np.array_equal(gq, qh) and np.array_equal(gb, bh) and np.array_equal(gs, sc)


@pytest.mark.parametrize("shape", ["clean", "crlf", "dirty", "hifi", "mixed", "wide"])
def test_stats_kilobase_reads(fqref, gpu, shape):
    """Reads of 2 - 5 kbp (and a mix with short ones): k_stats_oct counts them in passes of 256 columns, every pass in
    LDS; the caller's lmax may be smaller than the reads (overflow counters), larger, or no multiple of anything."""
    rng = np.random.default_rng({"clean": 1, "crlf": 2, "dirty": 3, "hifi": 4, "mixed": 5, "wide": 6}[shape])
    recs = []
    for i in range(260):
        n = int(rng.integers(2000, 5001))
        if shape == "mixed" and i % 3:
            n = int(rng.integers(0, 300))
        seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n).tobytes()
        qual = rng.integers(33, 75, n).astype(np.uint8).tobytes()
        if shape == "hifi":
            qual = rng.choice(np.frombuffer(b"~!5I", dtype=np.uint8), n).tobytes()   # '~' = Q93: beyond the 64 bins, inside k_stats_long's 128
        if shape == "wide":   # both edges of k_stats_long's 128-bin window '!' .. 0xA0, and bytes on either side of it
            qual = rng.choice(np.array([0x21, 0x60, 0x61, 0x7E, 0x7F, 0x80, 0xA0, 0xA1, 0xFF, 0x20, 0x0B, 0x00], dtype=np.uint8), n,
                              p=[.2, .1, .1, .3, .05, .05, .1, .04, .02, .02, .01, .01]).tobytes()
        if shape == "dirty" and i % 4 == 0 and n:
            sa = bytearray(seq)
            for k in rng.integers(0, n, 3):
                sa[int(k)] = int(rng.choice(list(b"Nnx.")))
            sa[n - 1] = ord("N") if i % 8 == 0 else sa[n - 1]     # an 'N' in the line's last column, far beyond the first pass
            seq = bytes(sa)
        e = b"\r\n" if (shape == "crlf" and i % 2 == 0) else b"\n"
        recs.append(b"@m%d/ccs" % i + e + seq + e + b"+" + e + qual + e)
    data = b"".join(recs)
    for lmax in (5000, 4097, 1000, 256, 257, 6000):
        r, qh, bh, sc = fqref.stats(data, lmax)
        s, gq, gb, gs = gpu.stats(data, lmax)
        assert (s.parse_status, s.n_records) == (r.status, r.n_records) == (0, 260)
        assert np.array_equal(gs, sc), (shape, lmax, gs, sc)
        assert np.array_equal(gb, bh), (shape, lmax, np.argwhere(gb != bh)[:5])
        assert np.array_equal(gq, qh), (shape, lmax, np.argwhere(gq != qh)[:5])


@pytest.mark.parametrize("shape", ["tail", "two_classes", "one_class", "few", "empty_lines"])
def test_stats_long_reads_of_many_lengths(fqref, gpu, shape):
    """k_stats_long's plan is made on the device from the reads' lengths (k_long_census / k_long_plan / k_long_scatter): records
    ordered by the number of 256-column blocks they reach, every column block's prefix cut into items of equal size.  Enough
    records for several items per column block, lengths with a long tail (classes 1 .. 36), two classes only, one class (no
    sorted copy), fewer records than one item holds, and reads of length 0 among long ones; CRLF lines and bytes outside the
    alphabets on the way.  (src/lib.rs:276-283 and src/records.rs:75-90 treat every record up to BUFSIZE alike.)"""
    rng = np.random.default_rng({"tail": 11, "two_classes": 12, "one_class": 13, "few": 14, "empty_lines": 15}[shape])
    n_rec = {"tail": 5000, "two_classes": 3000, "one_class": 3000, "few": 70, "empty_lines": 1500}[shape]
    recs = []
    for i in range(n_rec):
        if shape == "tail":
            n = int(min(9000, rng.geometric(1 / 900.0)))
        elif shape == "two_classes":
            n = 700 if i % 5 else 2100
        elif shape == "one_class":
            n = int(rng.integers(513, 769))
        elif shape == "few":
            n = int(rng.integers(1, 4000))
        else:
            n = 0 if i % 3 == 0 else int(rng.integers(1, 1500))
        seq = bytearray(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n).tobytes())
        qual = bytearray(rng.choice(np.frombuffer(b"~!5I", dtype=np.uint8), n).tobytes() if i % 2 else rng.integers(33, 75, n).astype(np.uint8).tobytes())
        if n and i % 7 == 0:
            seq[n - 1] = ord("N")                       # in the record's last column block
        if n > 300 and i % 11 == 0:
            seq[int(rng.integers(0, n))] = ord("x")
            qual[int(rng.integers(0, n))] = 0xF0         # outside the 128-bin window
        e = b"\r\n" if i % 13 == 0 else b"\n"
        recs.append(b"@r%d" % i + e + bytes(seq) + e + b"+" + e + bytes(qual) + e)
    data = b"".join(recs)
    for lmax in (9000, 1000, 600):
        r, qh, bh, sc = fqref.stats(data, lmax)
        s, gq, gb, gs = gpu.stats(data, lmax)
        assert (s.parse_status, s.n_records) == (r.status, r.n_records) == (0, n_rec)
        assert np.array_equal(gs, sc), (shape, lmax, gs, sc)
        assert np.array_equal(gb, bh), (shape, lmax, np.argwhere(gb != bh)[:5])
        assert np.array_equal(gq, qh), (shape, lmax, np.argwhere(gq != qh)[:5])


@pytest.mark.parametrize("case", ["short_reads_many_rows", "hint_too_small", "second_file_longer", "header_longer_than_reads",
                                  "longer_reads_further_down"])
def test_single_pass_rows_follow_the_reads(fqref, torch, pkg, case):
    """lmax is the caller's choice; the reference's closure has none (src/lib.rs:226-237).  The single pass sizes its rows by the
    reads (the first 64 KiB of the first input a context sees, then what its calls find), at most lmax: rows of 1000 over reads
    of 150 bases are one pass; a first window that holds only short reads in front of longer ones lists the longer lines (counted
    behind the pass, columns beyond its rows included) or gives the pass up — bit-exact either way — and the next call knows."""
    rng = np.random.default_rng({"short_reads_many_rows": 1, "hint_too_small": 2, "second_file_longer": 3, "header_longer_than_reads": 4,
                                 "longer_reads_further_down": 5}[case])
    def reads(n, lo, hi, hdr=8):
        out = []
        for i in range(n):
            L = int(rng.integers(lo, hi + 1))
            out.append(b"@" + bytes(rng.integers(97, 123, hdr).astype(np.uint8).tolist()) + b"\n" +
                       rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), L, p=[.2475, .2475, .2475, .2475, .01]).tobytes() + b"\n+\n" +
                       rng.integers(33, 75, L).astype(np.uint8).tobytes() + b"\n")
        return b"".join(out)
    gpu = Gpu(torch, pkg.Ctx(0), pkg)
    if case == "short_reads_many_rows":
        files = [(reads(30000, 150, 150), 1000, {1}), (reads(30000, 150, 150), 512, {1}), (reads(30000, 36, 150), 2000, {1})]
    elif case == "hint_too_small":
        # 800 reads of 50 bases fill the first 64 KiB; then a few of 300 (listed: route 2), then a file where most are (given up or listed)
        # (every file is another input: the context looks at each one's first 64 KiB itself — ADVICE r5 — so the third file's rows
        # are what ITS window shows: reads longer than that further down are listed, or — too many of them — the pass is given up)
        files = [(reads(800, 50, 50) + reads(20, 300, 300) + reads(20000, 50, 50), 1000, {2}),
                 (reads(800, 50, 50) + reads(20000, 50, 300), 1000, {0, 1, 2}), (reads(20000, 50, 300), 1000, {0, 1, 2})]
    elif case == "second_file_longer":
        # (a pass that is given up also makes the context skip its next attempt: the third file may go either way, the fourth may not;
        # kilobase reads fail the scan's fast path and with it the single pass, whose back-off the next statistics call counts
        # down — fqh_stats calls alone used to leave it where it was, for good — so the file after the next is one pass again)
        # what a context believes about the reads' length belongs to ONE input.  A 250 bp file behind a 100 bp one is
        # looked at itself and takes the single pass at once (until round 5 it was sized by the 100 bp file, listed every line,
        # gave the pass up and sent the NEXT call to two passes as well); kilobase reads take two passes without an attempt, and
        # the 100 bp file behind them is one pass again
        files = [(reads(20000, 100, 100), 700, {1}), (reads(20000, 250, 250), 700, {1}), (reads(20000, 250, 250), 700, {1}),
                 (reads(20000, 250, 250), 700, {1}), (reads(3000, 2000, 2500), 3000, {0}), (reads(20000, 100, 100), 700, {0, 1}),
                 (reads(20000, 100, 100), 700, {1}), (reads(20000, 100, 100), 700, {1})]
    elif case == "longer_reads_further_down":
        # the look takes four windows, a quarter of the input apart (round 6): half a megabyte of 50-base reads in front of 150-base
        # ones — a run trimmed harder at its start — is ONE pass sized by the longer reads; the later windows begin inside lines
        # and settle their line phase themselves ('@' / '+' residues), also with quality lines that begin with '@' or '+' and with
        # headers longer than the reads
        files = [(reads(5000, 50, 50) + reads(60000, 150, 150), 1000, {1}),
                 (reads(5000, 36, 36, hdr=180) + reads(40000, 120, 120, hdr=180), 300, {1}),
                 (reads(6000, 50, 50) + reads(30000, 250, 250), 250, {1})]
    else:
        files = [(reads(20000, 36, 36, hdr=200), 600, {1})]
    for data, lmax, routes in files:
        r, qh, bh, sc = fqref.stats(data, lmax)
        s, gq, gb, gs = gpu.stats(data, lmax)
        assert (s.parse_status, s.n_records) == (r.status, r.n_records) and r.status == 0
        assert np.array_equal(gs, sc), (case, lmax, gs, sc)
        assert np.array_equal(gq, qh) and np.array_equal(gb, bh), (case, lmax)
        assert gpu.ctx.last_stats_route() in routes, (case, lmax, gpu.ctx.last_stats_route(), routes)
    gpu.ctx.close()


def test_two_contexts_on_two_host_threads(fqref, torch, pkg):
    """A context is single-threaded, a process is not (parallel_each's workers, src/lib.rs:521-559, are threads of ONE process):
    two host threads with a context and a stream each, first calls at the same time — the process-wide bookkeeping behind the
    launches (LDS attributes per kernel and device, occupancy look-ups) is shared — on inputs that take different kernels."""
    import threading
    rng = np.random.default_rng(5)
    def reads(n, L):
        return b"".join(b"@r%d\n" % i + rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), L, p=[.2475, .2475, .2475, .2475, .01]).tobytes() + b"\n+\n" +
                        rng.integers(33, 75, L).astype(np.uint8).tobytes() + b"\n" for i in range(n))
    jobs = [(reads(20000, 150), 150), (reads(3000, 1200), 1200), (reads(20000, 300), 300), (reads(20000, 36), 36)]
    want = [fqref.stats(d, lmax) for d, lmax in jobs]
    errors = []
    def work(which):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                gpu = Gpu(torch, pkg.Ctx(0, stream=stream.cuda_stream), pkg)
                for rep in range(6):
                    for k in (which, which + 2):
                        (data, lmax), (r, qh, bh, sc) = jobs[k], want[k]
                        s, gq, gb, gs = gpu.stats(data, lmax)
                        assert (s.parse_status, s.n_records) == (r.status, r.n_records)
                        assert np.array_equal(gs, sc) and np.array_equal(gq, qh) and np.array_equal(gb, bh), (which, k, rep)
                gpu.ctx.close()
        except Exception as e:   # (an assertion in a thread is nobody's unless it is carried out)
            errors.append((which, repr(e)))
    threads = [threading.Thread(target=work, args=(w,)) for w in (0, 1)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_single_pass_backs_off_after_a_pass_it_gave_up(fqref, torch, pkg):
    """A file whose reads are soft-masked by the thousand (any byte may stand in seq(), src/records.rs:75-90) has more batches with
    a byte outside ACGTN than the single pass can dump: it gives the pass up (route 0: counted over the exact index, bit-exact) —
    and the context's next 1, 2, 4, .. statistics calls go straight to that route instead of paying for another attempt (an
    attempt leaves the scan's result on the fast path, a skipped one is the exact path's); a pass that commits forgets the
    back-off.  The closure of Parser::each (src/lib.rs:226-237) sees every record whatever the caller does with it: results
    never depend on the route."""
    rng = np.random.default_rng(31)
    nrec = 40000
    def reads(masked):
        recs = []
        for i in range(nrec):
            sq = bytearray(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), 150, p=[.2475, .2475, .2475, .2475, .01]).tobytes())
            if masked and i % 10 == 3:
                sq[20:60] = bytes(sq[20:60]).lower()
            recs.append(b"@r%d\n" % i + bytes(sq) + b"\n+\n" + rng.integers(33, 75, 150).astype(np.uint8).tobytes() + b"\n")
        return b"".join(recs)
    dirty, clean = reads(True), reads(False)
    gpu = Gpu(torch, pkg.Ctx(0), pkg)
    want = {id(dirty): fqref.stats(dirty, 150), id(clean): fqref.stats(clean, 150)}
    dev = {id(dirty): gpu.upload(dirty), id(clean): gpu.upload(clean)}   # two resident inputs (two buffers)
    seen = []

    def count(data):
        r, qh, bh, sc = want[id(data)]
        d, n = dev[id(data)]
        gq = torch.zeros(150 * 256, dtype=torch.int64, device=gpu.dev)
        gb = torch.zeros(150 * 8, dtype=torch.int64, device=gpu.dev)
        gs = torch.zeros(8, dtype=torch.int64, device=gpu.dev)
        s, _ = gpu.ctx.stats(d.data_ptr(), n, 150, gq.data_ptr(), gb.data_ptr(), gs.data_ptr())
        assert (s.parse_status, s.n_records) == (r.status, r.n_records) == (0, nrec)
        assert np.array_equal(gs.cpu().numpy().astype(np.uint64), sc)
        assert np.array_equal(gq.cpu().numpy().astype(np.uint64).reshape(150, 256), qh)
        assert np.array_equal(gb.cpu().numpy().astype(np.uint64).reshape(150, 8), bh)
        seen.append((gpu.ctx.last_stats_route(), bool(gpu.ctx.last_scan_fast())))

    for _ in range(7):
        count(dirty)
    # calls 1, 3 and 6 try and give up (back-off 1, 2, 4), calls 2, 4, 5 and 7 are skipped; call 8 on the SAME input would be
    # skipped as well (three more to go) ...
    assert seen[:7] == [(0, True), (0, False), (0, True), (0, False), (0, False), (0, True), (0, False)], seen
    # ... but the back-off belongs to the input that earned it: another input — the clean file, another buffer — is
    # looked at itself and takes the single pass at once; and back on the dirty one the count-down starts over
    count(clean)
    count(clean)
    assert seen[7:] == [(1, True), (1, True)], seen
    count(dirty)
    assert seen[9] == (0, True), seen
    gpu.ctx.close()


@pytest.mark.parametrize("huge", [1_000_000, 1_250_000])
def test_stats_one_huge_read_among_short_ones(fqref, gpu, huge):
    """A Buffer of 4 MiB (Parser::new takes any reader, src/lib.rs:208; the Buffer's size is the crate's constant, a caller's
    choice here) and one read of a million bases among reads of a few hundred: 3 907 column blocks — the planner's arrays hold
    4 096, k_long_census / k_long_plan / k_long_lists run with nearly all of them in use by ONE record — and 4 883, beyond
    what the planner holds: equal slices, no census."""
    rng = np.random.default_rng(huge)
    recs = []
    for i in range(60):
        n = huge if i == 17 else int(rng.integers(0, 900))
        seq = bytearray(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n).tobytes())
        qual = rng.choice(np.frombuffer(b"~!5I", dtype=np.uint8), n).tobytes()
        if n:
            seq[n - 1] = ord("N")
        e = b"\r\n" if i % 9 == 0 else b"\n"
        recs.append(b"@r%d\n" % i + bytes(seq) + e + b"+\n" + qual + e)
    data = b"".join(recs)
    B = 4 << 20
    for lmax in (2000, 700):
        r, qh, bh, sc = fqref.stats(data, lmax, bufsize=B)
        s, gq, gb, gs = gpu.stats(data, lmax, bufsize=B)
        assert (s.parse_status, s.n_records) == (r.status, r.n_records) == (0, 60)
        assert np.array_equal(gs, sc), (huge, lmax, gs, sc)
        assert np.array_equal(gb, bh) and np.array_equal(gq, qh), (huge, lmax)


@pytest.mark.parametrize("shape", ["fixed150", "fixed36", "ragged", "binned", "crlf", "dirty", "len4k"])
def test_stats_fast_path_shapes(fqref, torch, pkg, shape):
    """Multi-tile buffers (so that the whole-dword LDS path runs, not only the exact one): fixed and
    ragged read lengths, binned qualities (four distinct values: the worst case for LDS atomics
    keyed by bin), CRLF, sprinkled bytes outside the alphabet / quality window, and lengths that
    are / are not multiples of 4 -- against the oracle for row counts around every kernel variant."""
    rng = np.random.default_rng({"fixed150": 1, "fixed36": 2, "ragged": 3, "binned": 4, "crlf": 5, "dirty": 6,
                                 "len4k": 7}[shape])
    recs = []
    nrec = 6000 if shape != "fixed36" else 20000
    for i in range(nrec):
        if shape == "fixed150":
            n = 150
        elif shape == "fixed36":
            n = 36
        elif shape == "len4k":
            n = int(rng.choice([148, 152, 256, 260, 0, 1, 3, 4]))
        else:
            n = int(rng.integers(0, 301))
        seq = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), n, p=[.24, .25, .25, .25, .01]).tobytes()
        if shape == "binned":
            qual = rng.choice(np.frombuffer(b"#,:F", dtype=np.uint8), n).tobytes()
        else:
            qual = rng.integers(33, 75, n).astype(np.uint8).tobytes()
        if shape == "dirty" and i % 7 == 0 and n:
            b = bytearray(seq)
            b[int(rng.integers(0, n))] = int(rng.choice(list(b"acgtn*.-")))
            seq = bytes(b)
            q = bytearray(qual)
            q[int(rng.integers(0, n))] = int(rng.choice([32, 97, 126, 200, 13]))
            qual = bytes(q)
        e = b"\r\n" if (shape == "crlf" and i % 3 == 0) else b"\n"
        recs.append(b"@read%d" % i + e + seq + e + b"+" + e + qual + e)
    data = b"".join(recs)
    assert len(data) > 20 * 16384
    lmaxes = {"fixed150": (150, 149, 151, 160, 128, 64), "fixed36": (36, 40, 64), "ragged": (100, 101, 250, 256, 300),
              "binned": (150, 300), "crlf": (150, 257), "dirty": (64, 150, 300), "len4k": (148, 152, 256, 260)}[shape]
    gpu = Gpu(torch, pkg.Ctx(0), pkg)   # (a context of its own: the route is pinned, and list sizes / back-offs stick to a context)
    for lmax in lmaxes:
        gpu.ctx.set_single_pass(True)   # (forgets the back-off a pass that was given up leaves behind: every row count is tried)
        r, qh, bh, sc = fqref.stats(data, lmax)
        s, gq, gb, gs = gpu.stats(data, lmax)
        assert (s.parse_status, s.n_records) == (r.status, r.n_records) and r.n_records == nrec
        assert np.array_equal(gs, sc), (shape, lmax, gs, sc)
        assert np.array_equal(gq, qh) and np.array_equal(gb, bh), (shape, lmax)
        # the route: every row count takes the scan's own pass (1) — also rows fewer than the reads are long: the pass keeps the
        # rows the READS need and what it counts beyond lmax becomes the overflow counters —; 2 where lines hold bytes outside the
        # alphabets (counted behind it), 0 where those outnumber its dump area (one slot per 512 KiB)
        want = {1} if shape != "dirty" else {0, 2}
        assert gpu.ctx.last_stats_route() in want, (shape, lmax, gpu.ctx.last_stats_route())
    gpu.ctx.close()


def test_synth_generator_and_medium_parity(fqref, gpu, torch):
    """64 MiB of the synthetic 150 bp workload: generator identical on CPU and GPU; offsets, count
    and histograms bit-exact against the oracle."""
    n = 64 * 1024 * 1024
    d = torch.empty(n, dtype=torch.uint8, device=gpu.dev)
    gpu.ctx.synth_fill(d.data_ptr(), 0, n)
    host = d.cpu().numpy()
    assert np.array_equal(host, fqref.synth(0, n))
    res, off = fqref.offsets(host)
    s, c, offs = gpu.scan_dev(d, n)
    assert (s.parse_status, s.n_records) == (res.status, res.n_records)  # ends mid-record: truncated
    assert np.array_equal(offs[:-1], off)
    nrec = n // 330
    full = nrec * 330
    s, c, offs = gpu.scan_dev(d, full)
    assert s.parse_status == gpu.pkg.OK and s.n_records == nrec
    assert np.array_equal(offs, np.arange(nrec + 1, dtype=np.uint64) * 330)
    qh = torch.zeros(150 * 256, dtype=torch.int64, device=gpu.dev)
    bh = torch.zeros(150 * 8, dtype=torch.int64, device=gpu.dev)
    sc = torch.zeros(8, dtype=torch.int64, device=gpu.dev)
    gpu.ctx.stats(d.data_ptr(), full, 150, qh.data_ptr(), bh.data_ptr(), sc.data_ptr())
    r, oq, ob, osc = fqref.stats(host[:full], 150)
    assert np.array_equal(qh.cpu().numpy().astype(np.uint64).reshape(150, 256), oq)
    assert np.array_equal(bh.cpu().numpy().astype(np.uint64).reshape(150, 8), ob)
    assert np.array_equal(sc.cpu().numpy().astype(np.uint64), osc)


@pytest.mark.parametrize("seed", range(4))
def test_chunked_scan_with_carry(fqref, gpu, pkg, seed):
    """Arbitrary byte cuts (the multi-GPU shard boundaries of SURVEY §8e): chaining fqh_scan with the
    carry reproduces the whole-file scan exactly."""
    rng = np.random.default_rng(40 + seed)
    data = fuzzgen.valid_file(rng, 400, maxlen=60)
    if seed == 3:
        data = fuzzgen.mutate(rng, data, 1)
    res, idx = fqref.index(data, bufsize=1 << 22)
    gpu.ctx.set_bufsize(0)
    ncut = int(rng.integers(1, 6))
    cuts = sorted(set([0, len(data)] + [int(x) for x in rng.integers(0, len(data) + 1, ncut)]))
    carry = None
    starts = []
    total = 0
    status = pkg.OK
    for a, b in zip(cuts[:-1], cuts[1:]):
        # device sub-buffers must be 16-byte aligned: upload each chunk separately
        s, carry, offs = gpu.scan(data[a:b], is_final=(b == len(data)), carry=carry, bufsize=0)
        if not starts:
            starts.append(int(offs[0]))
        starts += [int(x) for x in offs[1: s.n_records + 1]]
        total += s.n_records
        assert carry.base_offset == b
        if s.parse_status != pkg.OK:
            status = s.parse_status
            break
    assert status == res.status
    assert total == res.n_records
    assert starts[: res.n_records] == [int(x) for x in idx[:, 0]]


@pytest.mark.parametrize("bufsize", [256, 69632])
def test_chunked_scan_judges_too_long_on_file_offsets(fqref, gpu, pkg, bufsize):
    """"Fastq record is too long" (src/lib.rs:278-283) depends on the record's file offset mod 16 (csrc/replay.h, fqh::TooLong), so
    fqh_scan chained over chunks with the carry gives Parser::each's verdict for records of BUFSIZE - 17 .. BUFSIZE + 2 bytes:
    complete ones, one that straddles a cut, the record in progress at the end of a chunk, a truncated one at the end."""
    rng = np.random.default_rng(bufsize)

    def rec(total):
        body = total - 6
        sl = int(rng.integers(0, body // 2 + 1))
        return b"@" + b"h" * (body - 2 * sl) + b"\n" + b"A" * sl + b"\n+\n" + b"I" * sl + b"\n"

    seen = set()
    for trial in range(10 if bufsize > 1000 else 40):
        parts = []
        for i in range(int(rng.integers(3, 40))):
            L = int(rng.integers(bufsize - 17, bufsize + 3)) if rng.random() < 0.12 else int(rng.integers(6, bufsize // 2))
            parts.append(rec(max(6, L)))
        data = b"".join(parts)
        if trial % 5 == 4:
            data = data[: len(data) - int(rng.integers(1, min(len(data) - 1, bufsize)))]
        res = fqref.count(data, bufsize=bufsize)
        cuts = sorted(set([0, len(data)] + [int(x) for x in rng.integers(0, len(data) + 1, int(rng.integers(1, 5)))]))
        carry, total, status = None, 0, pkg.OK
        for a, b in zip(cuts[:-1], cuts[1:]):
            s, carry, offs = gpu.scan(data[a:b], is_final=(b == len(data)), carry=carry, bufsize=bufsize)
            total += s.n_records
            if s.parse_status != pkg.OK:
                status = s.parse_status
                break
        assert (status, total) == (res.status, res.n_records), (trial, cuts, status, total, res.status, res.n_records)
        seen.add(res.status)
    assert pkg.E_TOO_LONG in seen and pkg.OK in seen


def test_count_only_mode(fqref, gpu):
    rng = np.random.default_rng(2)
    data = fuzzgen.valid_file(rng, 300)
    s, c, _ = gpu.scan(data, want_offsets=False)
    r = fqref.count(data)
    assert (s.parse_status, s.n_records, s.bytes_consumed) == (r.status, r.n_records, r.bytes_consumed)
    bad = fuzzgen.mutate(rng, data, 2)
    s, c, _ = gpu.scan(bad, want_offsets=False)
    r = fqref.count(bad)
    assert (s.parse_status, s.n_records) == (r.status, r.n_records)
    if r.n_records:
        assert s.bytes_consumed == r.bytes_consumed


@pytest.mark.parametrize("seed", range(3))
def test_sharded_two_stage_protocol(fqref, torch, pkg, seed):
    """The multi-GPU protocol of bench.py on one GPU: every shard is byte-scanned with a zero carry,
    the 7-word summaries are folded with fqh_carry_combine, and fqh_rescan_launch redoes only the
    emit step with the true carry.  Result == whole-file oracle."""
    rng = np.random.default_rng(70 + seed)
    data = fuzzgen.valid_file(rng, 600, maxlen=80)
    res, idx = fqref.index(data, bufsize=1 << 22)
    nsh = 4
    cuts = [0] + sorted(int(x) for x in rng.integers(1, len(data), nsh - 1)) + [len(data)]
    dev = torch.device("cuda:0")
    ctxs, bufs, sums = [], [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        c = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream, bufsize=0)
        n = b - a
        d = torch.empty(max(n, 16), dtype=torch.uint8, device=dev)
        if n:
            d[:n].copy_(torch.from_numpy(np.frombuffer(data[a:b], dtype=np.uint8).copy()))
        if seed == 0:
            s0, c0, _ = c.scan(d.data_ptr(), n, False, None, None, 0)
            summ = (n, s0.n_newlines, s0.n_line_starts, list(c0.back))
        else:
            nn, ns, back0 = c.shard_prescan(d.data_ptr(), n)  # same numbers without the emit pass
            summ = (n, nn, ns, back0)
        ctxs.append(c); bufs.append((d, n)); sums.append(summ)
    starts, total = [], 0
    carry = None
    for i, c in enumerate(ctxs):
        d, n = bufs[i]
        cap = n // 6 + 3
        rs = torch.zeros(cap, dtype=torch.int64, device=dev)
        c.rescan_launch(i == nsh - 1, carry, rs.data_ptr(), cap)
        s, cout, st = c.scan_finish()
        assert s.parse_status == pkg.OK
        offs = rs.cpu().numpy()
        if not starts:
            starts.append(int(offs[0]))
        starts += [int(x) for x in offs[1: s.n_records + 1]]
        total += s.n_records
        carry = pkg.carry_combine(carry, *sums[i])
        assert (carry.base_offset, carry.nl_count, list(carry.back)) == \
            (cout.base_offset, cout.nl_count, list(cout.back))
        c.close()
    assert total == res.n_records
    assert starts[: res.n_records] == [int(x) for x in idx[:, 0]]


@pytest.mark.parametrize("kind", ["reads150", "ragged", "short_shard", "error_in_shard", "long_reads"])
def test_sharded_protocol_on_the_device(fqref, torch, pkg, kind):
    """The same protocol with the exchange on the device: fqh_shard_prescan_launch writes a rank's 8 words to device
    memory, (here: straight into its row of the gathered array; across GPUs: an all-gather), fqh_shard_rescan_launch folds
    the rows in front of the rank and emits under that carry — no host hop in between.  Result == whole-file oracle;
    a shard that cannot keep the fast path makes every rank's finish return E_AGAIN (the host recipe then runs)."""
    rng = np.random.default_rng(5)
    alph = np.frombuffer(b"ACGT", dtype=np.uint8)
    recs = []
    for i in range(9000 if kind != "long_reads" else 400):
        L = 150 if kind in ("reads150", "error_in_shard") else int(rng.integers(1, 200))
        if kind == "long_reads":
            L = 9000   # fewer than eight line starts per 16 KiB tile: not something the fast path proves
        recs.append(b"@r%d\n" % i + rng.choice(alph, L).tobytes() + b"\n+\n" + rng.integers(33, 74, L).astype(np.uint8).tobytes() + b"\n")
    data = bytearray(b"".join(recs))
    n = len(data)
    nsh = 4
    cuts = [0] + sorted(int(x) for x in rng.integers(n // 8, n - n // 8, nsh - 1)) + [n]
    if kind == "short_shard":
        cuts[2] = cuts[1] + 37          # a shard inside one line
    if kind == "error_in_shard":
        k = data.index(b"\n+\n", cuts[2] + 5000)
        data[k + 1] = ord("-")          # in the third shard
    data = bytes(data)
    res, idx = fqref.index(data, bufsize=1 << 22)
    dev = torch.device("cuda:0")
    W = pkg.SHARD_WORDS
    stream = torch.cuda.Stream(device=dev)   # ONE stream for all the "ranks": nothing else orders their kernels here
    torch.cuda.set_stream(stream)
    all_words = torch.zeros(nsh * W, dtype=torch.int64, device=dev)
    counts = torch.zeros(nsh * 2, dtype=torch.int64, device=dev)
    ctxs, bufs, outs = [], [], []
    for r, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        c = pkg.Ctx(0, stream=stream.cuda_stream, bufsize=0)
        d = torch.empty(max(b - a, 16), dtype=torch.uint8, device=dev)
        d[: b - a].copy_(torch.from_numpy(np.frombuffer(data[a:b], dtype=np.uint8).copy()))
        c.shard_prescan_launch(d.data_ptr(), b - a, all_words[r * W:].data_ptr())
        ctxs.append(c); bufs.append((d, b - a))
    for r, c in enumerate(ctxs):
        cap = bufs[r][1] // 6 + 3
        rs = torch.zeros(cap, dtype=torch.int64, device=dev)
        c.shard_rescan_launch(r == nsh - 1, all_words.data_ptr(), nsh, r, rs.data_ptr(), cap, counts[2 * r:].data_ptr())
        outs.append(rs)
    again, starts, total, status = 0, [], 0, pkg.OK
    for r, c in enumerate(ctxs):
        try:
            s, cout, st = c.scan_finish()
        except pkg.FqhError as e:
            assert e.status == pkg.E_AGAIN
            again += 1
            continue
        offs = outs[r].cpu().numpy()
        if status == pkg.OK:
            if not starts:
                starts.append(int(offs[0]))
            starts += [int(x) for x in offs[1: s.n_records + 1]]
            total += s.n_records
            if s.parse_status != pkg.OK:
                status = s.parse_status
                assert (s.err_record, s.err_offset) == (res.err_record, res.err_offset)
        assert int(counts[2 * r]) == s.n_records or s.parse_status != pkg.OK
    if kind in ("long_reads", "short_shard", "error_in_shard"):
        # (a shard of 37 bytes has no four line starts to show; a tile with a broken record settles no alignment)
        assert again == nsh       # every rank learns it from the gathered words
        carry, starts, total, status = None, [], 0, pkg.OK
        for r, c in enumerate(ctxs):   # the host recipe, on the same contexts
            nn, ns, back0 = c.shard_prescan(bufs[r][0].data_ptr(), bufs[r][1])
            c.rescan_launch(r == nsh - 1, carry, outs[r].data_ptr(), outs[r].numel())
            s, cout, st = c.scan_finish()
            carry = pkg.carry_combine(carry, bufs[r][1], nn, ns, back0)
            if status == pkg.OK:
                offs = outs[r].cpu().numpy()
                if not starts:
                    starts.append(int(offs[0]))
                starts += [int(x) for x in offs[1: s.n_records + 1]]
                total += s.n_records
                if s.parse_status != pkg.OK:
                    status = s.parse_status
                    assert (s.err_record, s.err_offset) == (res.n_records, res.bytes_consumed)  # the failing record: the one after the last good one
        assert (status, total) == (res.status, res.n_records)
        assert starts[: res.n_records] == [int(x) for x in idx[: res.n_records, 0]]
    else:
        assert again == 0
        assert (status, total) == (res.status, res.n_records) == (pkg.OK, len(recs))
        assert starts[: res.n_records] == [int(x) for x in idx[: res.n_records, 0]]
        assert int(counts[1::2].sum()) == 0 and int(counts[0::2].sum()) == res.n_records
    for c in ctxs:
        c.close()
    torch.cuda.set_stream(torch.cuda.default_stream(dev))


@pytest.mark.parametrize("seed", range(3))
def test_sharded_histograms_with_tail_exchange(fqref, torch, pkg, seed):
    """bench.py --shard-stats on one GPU: byte-range shards cut anywhere; every shard gets the tail of the
    previous one in front of its buffer (what the all_gather of tails delivers), is scanned with the
    combined carry and histogrammed with fqh_stats_launch_lead; the sum over the shards (the all_reduce)
    equals the oracle's histograms of the whole file, scalars included."""
    rng = np.random.default_rng(170 + seed)
    data = fuzzgen.valid_file(rng, 2500, maxlen=200, crlf=(seed == 2))
    lmax = 200
    r, qh, bh, sc = fqref.stats(data, lmax)
    nsh = 4
    lead = 2 * pkg.BUFSIZE
    cuts = [0] + sorted(int(x) for x in rng.integers(1, len(data), nsh - 1)) + [len(data)]
    dev = torch.device("cuda:0")
    gq = torch.zeros(lmax * 256, dtype=torch.int64, device=dev)
    gb = torch.zeros(lmax * 8, dtype=torch.int64, device=dev)
    gs = torch.zeros(8, dtype=torch.int64, device=dev)
    carry, total = None, 0
    for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        n = b - a
        store = torch.zeros(lead + max(n, 16) + 16, dtype=torch.uint8, device=dev)
        have = min(a, lead)  # what the previous shards can deliver (short shards: less than a full tail)
        chunk = np.frombuffer(data[a - have: b], dtype=np.uint8).copy()
        store[lead - have: lead + n].copy_(torch.from_numpy(chunk))
        c = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream, bufsize=0)
        d_ptr = store.data_ptr() + lead
        nn, ns, back0 = c.shard_prescan(d_ptr, n)
        c.rescan_launch(i == nsh - 1, carry, None, 0)
        s, cout, st = c.scan_finish()
        assert s.parse_status == pkg.OK
        c.stats_launch_lead(d_ptr, n, have, lmax, gq.data_ptr(), gb.data_ptr(), gs.data_ptr(),
                            is_final=(i == nsh - 1), carry=carry)
        s2, _ = c.stats_finish()
        assert s2.n_records == s.n_records
        total += s.n_records
        carry = pkg.carry_combine(carry, n, nn, ns, back0)
        c.close()
    assert total == r.n_records
    assert np.array_equal(gs.cpu().numpy().astype(np.uint64), sc)
    assert np.array_equal(gq.cpu().numpy().astype(np.uint64).reshape(lmax, 256), qh)
    assert np.array_equal(gb.cpu().numpy().astype(np.uint64).reshape(lmax, 8), bh)


@pytest.mark.parametrize("shape", ["plain_sep", "id_sep"])
def test_fast_path_alignment_from_the_first_entries(fqref, torch, pkg, shape):
    """k_index_fast takes a tile's alignment from its first four entries (the one that starts with '@' and has a '+' two
    entries on) and checks the tile's records under it.  Every record here has a quality line that starts with '@' and a
    sequence line that starts with '+' — the parser looks at neither (src/records.rs:141,155 test the header's and the
    separator's first byte only) — so the first entries offer two candidates in half of the tiles: `plain_sep` must fall back
    to the windows under all four alignments and KEEP the fast path; `id_sep` repeats the id behind the '+', which makes the
    wrong alignment consistent as well ('@id' and '+id' have one length): the tile cannot be settled, the scan reruns on
    the exact path.  Offsets and counts equal the oracle's either way."""
    rng = np.random.default_rng(31)
    recs = []
    for i in range(6000):
        n = int(rng.integers(60, 160))
        seq = b"+" + bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n - 1).tolist())
        qual = b"@" + bytes(rng.integers(33, 75, n - 1).astype(np.uint8).tolist())
        rid = b"r%d" % i
        recs.append(b"@" + rid + b"\n" + seq + b"\n+" + (rid if shape == "id_sep" else b"") + b"\n" + qual + b"\n")
    data = b"".join(recs)
    dev = torch.device("cuda:0")
    ctx = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    d = torch.empty(len(data) + 16, dtype=torch.uint8, device=dev)
    d[: len(data)].copy_(torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()))
    rs = torch.zeros(len(recs) + 1, dtype=torch.int64, device=dev)
    s, c, st = ctx.scan(d.data_ptr(), len(data), True, None, rs.data_ptr(), len(recs) + 1)
    res, idx = fqref.index(data)
    assert (s.parse_status, s.n_records) == (res.status, res.n_records) == (fqref.OK, len(recs))
    assert np.array_equal(rs.cpu().numpy().astype(np.uint64)[: res.n_records], idx[:, 0])
    assert ctx.last_scan_fast() == (shape == "plain_sep")
    ctx.close()


def test_fast_path_is_taken_and_falls_back_exactly(fqref, torch, pkg):
    """The fast path (record starts + tile edges only) must (a) really run on valid
    multi-tile input and (b) hand every input it cannot prove valid to the exact path: same status,
    count, offsets and error record as the oracle either way, with the fast path re-enabled per input."""
    rng = np.random.default_rng(909)
    dev = torch.device("cuda:0")
    ctx = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    taken = fell = 0
    for trial in range(60):
        nrec = int(rng.integers(300, 3000))
        data = fuzzgen.valid_file(rng, nrec, seqlen=int(rng.choice([20, 100, 150, 300])), crlf=bool(trial % 7 == 0))
        kind = trial % 4
        if kind == 1:
            data = fuzzgen.mutate(rng, data, 1)
        elif kind == 2:   # damage exactly at a tile boundary region
            b = bytearray(data)
            pos = min(len(b) - 1, 16384 * int(rng.integers(1, max(2, len(b) // 16384))) + int(rng.integers(-3, 4)))
            b[pos] = int(rng.choice(list(b"\n@+x")))
            data = bytes(b)
        elif kind == 3:
            data = data[: len(data) - int(rng.integers(0, 40))]
        ctx.set_spec(True)
        d = torch.empty(max(len(data), 16), dtype=torch.uint8, device=dev)
        d[: len(data)].copy_(torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()))
        cap = len(data) // 6 + 3
        rs = torch.zeros(cap, dtype=torch.int64, device=dev)
        s, c, st = ctx.scan(d.data_ptr(), len(data), True, None, rs.data_ptr(), cap)
        res, idx = fqref.index(data)
        assert (s.parse_status, s.n_records) == (res.status, res.n_records), (trial, kind)
        assert np.array_equal(rs.cpu().numpy().astype(np.uint64)[: res.n_records], idx[:, 0])
        if ctx.last_scan_fast():
            taken += 1
            assert res.status == fqref.OK
        else:
            fell += 1
        if res.status != fqref.OK:
            assert s.err_record == res.n_records
    assert taken >= 10 and fell >= 10, (taken, fell)
    ctx.close()


def test_fast_path_backoff_after_failure(fqref, torch, pkg):
    """A file the fast path cannot prove valid sends the context to the exact path for 1, 2, 4 ... later
    scans (not for good): results are oracle-exact on every call either way."""
    rng = np.random.default_rng(31)
    dev = torch.device("cuda:0")
    ctx = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    good = fuzzgen.valid_file(rng, 3000, seqlen=150)
    bad = fuzzgen.mutate(rng, good, 1)
    while fqref.count(bad).status == fqref.OK:
        bad = fuzzgen.mutate(rng, good, 1)

    def run(data):
        d = torch.empty(len(data) + 16, dtype=torch.uint8, device=dev)
        d[: len(data)].copy_(torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()))
        s, c, st = ctx.scan(d.data_ptr(), len(data), True, None, None, 0)
        r = fqref.count(data)
        assert (s.parse_status, s.n_records) == (r.status, r.n_records)
        return ctx.last_scan_fast()

    assert run(good) is True
    assert run(bad) is False                              # fails -> 1 scan of back-off
    assert [run(good) for _ in range(3)] == [False, True, True]
    assert run(bad) is False                              # back-off starts again at 1 after a success
    assert run(bad) is False                              # (this one is the back-off scan itself)
    assert run(bad) is False                              # second failure in a row -> 2 scans
    assert [run(good) for _ in range(4)] == [False, False, True, True]
    ctx.close()


@pytest.mark.parametrize("shape", ["ragged", "short", "tiny", "micro", "mixed_crlf"])
def test_quarter_gib_ragged_reads_offsets_and_histograms(fqref, torch, pkg, shape):
    """256 MiB of reads with ragged lengths (a ~3 MiB random block repeated; its size is not a multiple of
    the 16 KiB tile, so every repetition meets the tile grid differently): 16 384 tiles through the fast
    path (second lines for short reads included), offsets and histograms bit-exact against the oracle."""
    rng = np.random.default_rng({"ragged": 11, "short": 12, "tiny": 14, "micro": 15, "mixed_crlf": 13}[shape])
    if shape == "ragged":
        block = fuzzgen.valid_file(rng, 12000, maxlen=250)
    elif shape == "short":   # ~105 records per tile: the second line of the fast path's tile record
        block = fuzzgen.valid_file(rng, 25000, maxlen=140)
    elif shape == "tiny":    # ~870 line starts per tile: record starts beyond the two lines spill into the list area
        block = fuzzgen.valid_file(rng, 40000, maxlen=60)
    elif shape == "micro":   # more than 1024 line starts per tile: the fast path declines, the exact path answers
        block = fuzzgen.valid_file(rng, 80000, maxlen=20)
    else:
        block = b"".join(fuzzgen.valid_file(rng, 50, maxlen=200, crlf=bool(i & 1)) for i in range(200))
    reps = (256 << 20) // len(block)
    dev = torch.device("cuda:0")
    hb = torch.from_numpy(np.frombuffer(block, dtype=np.uint8).copy()).to(dev)
    d = torch.cat([hb.repeat(reps), torch.zeros(16, dtype=torch.uint8, device=dev)])
    n = reps * len(block)
    host = d[:n].cpu().numpy()
    res, off = fqref.offsets(host)
    assert res.status == fqref.OK
    ctx = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    rs = torch.zeros(res.n_records + 1, dtype=torch.int64, device=dev)
    s, c, st = ctx.scan(d.data_ptr(), n, True, None, rs.data_ptr(), res.n_records + 1)
    assert (s.parse_status, s.n_records) == (res.status, res.n_records)
    assert ctx.last_scan_fast() == (shape != "micro")
    assert np.array_equal(rs.cpu().numpy().astype(np.uint64)[:-1], off)
    lmax = 250
    r, qh, bh, sc = fqref.stats(host, lmax)
    gq = torch.zeros(lmax * 256, dtype=torch.int64, device=dev)
    gb = torch.zeros(lmax * 8, dtype=torch.int64, device=dev)
    gs = torch.zeros(8, dtype=torch.int64, device=dev)
    ctx.stats(d.data_ptr(), n, lmax, gq.data_ptr(), gb.data_ptr(), gs.data_ptr())
    assert np.array_equal(gs.cpu().numpy().astype(np.uint64), sc)
    assert np.array_equal(gq.cpu().numpy().astype(np.uint64).reshape(lmax, 256), qh)
    assert np.array_equal(gb.cpu().numpy().astype(np.uint64).reshape(lmax, 8), bh)
    ctx.close()


def test_full_size_16gib_properties(fqref, gpu, torch):
    """BASELINE.json's full size (configs[1] and [2]: 16 GiB of the synthetic 150 bp workload, HBM-resident)
    through properties that need no 16 GiB oracle run: every record offset (the generator's records
    are 330 bytes), the count, the whole-file histograms as the sum of the histograms of 32 record-
    aligned parts (linearity), their totals, and one part bit-exact against the oracle."""
    free, _ = torch.cuda.mem_get_info()
    if free < 20 * 2**30:
        pytest.skip("needs 20 GiB of device memory")
    nrec = (16 * 2**30) // 330
    n = nrec * 330
    d = torch.empty(n, dtype=torch.uint8, device=gpu.dev)
    gpu.ctx.synth_fill(d.data_ptr(), 0, n)
    gpu.ctx.set_bufsize(gpu.pkg.BUFSIZE)
    rs = torch.empty(nrec + 8, dtype=torch.int64, device=gpu.dev)
    s, c, st = gpu.ctx.scan(d.data_ptr(), n, True, None, rs.data_ptr(), nrec + 8)
    assert st == gpu.pkg.OK and s.parse_status == gpu.pkg.OK and s.n_records == nrec and s.bytes_consumed == n
    want = torch.arange(nrec + 1, dtype=torch.int64, device=gpu.dev) * 330
    assert bool(torch.equal(rs[: nrec + 1], want))
    del want, rs

    def hist(ptr, length):
        qh = torch.zeros(150 * 256, dtype=torch.int64, device=gpu.dev)
        bh = torch.zeros(150 * 8, dtype=torch.int64, device=gpu.dev)
        sc = torch.zeros(8, dtype=torch.int64, device=gpu.dev)
        s2, _ = gpu.ctx.stats(ptr, length, 150, qh.data_ptr(), bh.data_ptr(), sc.data_ptr())
        assert s2.parse_status == gpu.pkg.OK
        return qh, bh, sc

    qh, bh, sc = hist(d.data_ptr(), n)
    assert int(sc[0]) == nrec and int(sc[1]) == nrec * 150 and int(sc[2]) == nrec * 150
    assert int(qh.sum()) == nrec * 150 and int(bh.sum()) == nrec * 150
    assert bool((qh.view(150, 256).sum(1) == nrec).all()) and bool((bh.view(150, 8).sum(1) == nrec).all())
    parts = 32
    per = ((nrec + parts - 1) // parts + 7) // 8 * 8  # 8 records = 2640 bytes: the parts stay 16-byte aligned (fqh_stats)
    tq, tb, ts = torch.zeros_like(qh), torch.zeros_like(bh), torch.zeros_like(sc)
    for p in range(parts):
        r0, r1 = p * per, min(nrec, (p + 1) * per)
        q2, b2, s2 = hist(d.data_ptr() + r0 * 330, (r1 - r0) * 330)
        tq += q2
        tb += b2
        ts += s2
        if p == 17:  # one part against the oracle (the generator is a pure function of the byte offset)
            m = 64 * 2**20 // 330 * 330
            host = d[r0 * 330: r0 * 330 + m].cpu().numpy()
            assert np.array_equal(host, fqref.synth(r0 * 330, m))
            q3, b3, s3 = hist(d.data_ptr() + r0 * 330, m)
            r, oq, ob, osc = fqref.stats(host, 150)
            assert np.array_equal(q3.cpu().numpy().astype(np.uint64).reshape(150, 256), oq)
            assert np.array_equal(b3.cpu().numpy().astype(np.uint64).reshape(150, 8), ob)
            assert np.array_equal(s3.cpu().numpy().astype(np.uint64), osc)
    assert bool(torch.equal(tq, qh)) and bool(torch.equal(tb, bh)) and bool(torch.equal(ts, sc))


def test_fast_emit_routes(fqref, torch, pkg):
    """k_emit_fast has a streamlined per-tile loop (every tile of a 64-tile group keeps its record starts in its one line,
    all within the caller's capacity, no Buffer limit below two tiles) and a generic one for everything else.  Files that
    mix both kinds of groups, a capacity that ends inside a tile, and a small Buffer limit must give the oracle's
    offsets, counts, maximum record length and error either way."""
    rng = np.random.default_rng(4242)
    dev = torch.device("cuda:0")
    ctx = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)

    def upload(data):
        d = torch.empty(len(data) + 16, dtype=torch.uint8, device=dev)
        d[: len(data)].copy_(torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()))
        return d

    long_part = fuzzgen.valid_file(rng, 9000, seqlen=150)       # ~50 records per tile: the streamlined loop
    short_part = fuzzgen.valid_file(rng, 30000, seqlen=40)      # ~130 per tile: second line and list area
    mid_part = fuzzgen.valid_file(rng, 20000, seqlen=100)       # ~65 per tile: second line
    for name, data in (("long", long_part * 3), ("long+short", long_part * 2 + short_part + long_part * 2),
                       ("mid+long", mid_part + long_part * 3), ("short+mid+long", short_part + mid_part + long_part)):
        res, idx = fqref.index(data)
        assert res.status == fqref.OK
        starts = idx[:, 0]
        maxlen = int(np.max(np.diff(np.concatenate([starts, [len(data)]]).astype(np.int64))))
        d = upload(data)
        for cap_kind in ("full", "exact", "short"):
            cap = {"full": res.n_records + 64, "exact": res.n_records + 1, "short": res.n_records // 2 + 7}[cap_kind]
            rs = torch.full((cap + 8,), -1, dtype=torch.int64, device=dev)
            ctx.set_spec(True)
            ctx.set_bufsize(pkg.BUFSIZE)
            s, c, st = ctx.scan(d.data_ptr(), len(data), True, None, rs.data_ptr(), cap)
            assert (s.parse_status, s.n_records) == (fqref.OK, res.n_records), (name, cap_kind)
            assert st == (pkg.E_CAPACITY if cap_kind == "short" else pkg.OK), (name, cap_kind, st)
            got = rs.cpu().numpy()
            nchk = min(cap, res.n_records)
            assert np.array_equal(got[:nchk].astype(np.uint64), starts[:nchk]), (name, cap_kind)
            assert np.all(got[cap:] == -1), (name, cap_kind)        # nothing behind the capacity
            if cap_kind != "short":
                assert s.max_record_len == maxlen, (name, cap_kind)
                assert ctx.last_scan_fast(), (name, cap_kind)
        # a Buffer limit below two tiles: the generic loop keeps the per-record "too long" test
        lim = (maxlen + 15) // 16 * 16   # (a multiple of 16) the longest record trips the limit (src/lib.rs:276-283)
        ctx.set_spec(True)
        ctx.set_bufsize(lim)
        s, c, st = ctx.scan(d.data_ptr(), len(data), True, None, None, 0)
        r = fqref.count(data, bufsize=lim)
        assert (s.parse_status, s.n_records) == (r.status, r.n_records), (name, lim)
        ctx.set_bufsize(lim + 4096)
        ctx.set_spec(True)
        s, c, st = ctx.scan(d.data_ptr(), len(data), True, None, None, 0)
        r = fqref.count(data, bufsize=lim + 4096)
        assert (s.parse_status, s.n_records) == (r.status, r.n_records) == (fqref.OK, res.n_records)
    ctx.set_bufsize(pkg.BUFSIZE)
    ctx.close()


def test_read_length_histogram(fqref, gpu, torch, ctx):
    """fqh_len_hist: the read-length histogram (optional len_hist[len(seq())]) out of the base histogram the
    statistics call left behind, against the lengths of the oracle's records: ragged reads, empty reads, CRLF, reads longer
    than lmax (lumped in the last slot), and a file with a parse error (records before the error only)."""
    rng = np.random.default_rng(808)
    dev = torch.device("cuda:0")
    good = b"".join(fuzzgen.valid_record(rng, i, seqlen=int(rng.choice([0, 1, 7, 36, 100, 149, 150, 151, 200])),
                                         crlf=bool(i % 11 == 0)) for i in range(6000))
    bad = fuzzgen.mutate(rng, good, 1)
    while fqref.count(bad).status == fqref.OK:
        bad = fuzzgen.mutate(rng, good, 1)
    for data in (good, bad, good[:0]):
        for lmax in (150, 152, 256, 64):
            d, n = gpu.upload(data)
            qh = torch.zeros(lmax * 256, dtype=torch.int64, device=dev)
            bh = torch.zeros(lmax * 8, dtype=torch.int64, device=dev)
            sc = torch.zeros(8, dtype=torch.int64, device=dev)
            lh = torch.full((lmax + 1,), 5, dtype=torch.int64, device=dev)   # ADDED to
            ctx.set_spec(True)
            s, c = ctx.stats(d.data_ptr(), n, lmax, qh.data_ptr(), bh.data_ptr(), sc.data_ptr())
            ctx.len_hist(bh.data_ptr(), sc.data_ptr(), lmax, lh.data_ptr())
            res, idx = fqref.index(data)
            a = np.frombuffer(data, dtype=np.uint8)
            want = np.zeros(lmax + 1, dtype=np.int64)
            for k in range(res.n_records):
                start, head, seq = int(idx[k, 0]), int(idx[k, 1]), int(idx[k, 2])   # newline offsets relative to the record
                ln = seq - head - 1
                if ln and a[start + seq - 1] == 13:
                    ln -= 1                                                          # trim_winline, src/records.rs:66-73
                want[min(ln, lmax)] += 1
            assert s.n_records == res.n_records
            assert np.array_equal(lh.cpu().numpy() - 5, want), (len(data), lmax)


def test_workspace_placement_is_invisible(fqref, torch, pkg):
    """FQH_OPT_PLACE_TRIES: the first scan of 1 GiB or more on a context allocates several candidates of the per-tile line
    buffer, times the index kernel with each on the head of the caller's input and keeps the fastest.  Nothing but speed may
    depend on it: the same 1.25 GiB input through a context that places (4 tries), one that does not (0), and a second call
    on each, must give identical offsets, and the oracle's on a sample."""
    dev = torch.device("cuda:0")
    n = (5 << 28) // 330 * 330
    buf = torch.empty(n + 16, dtype=torch.uint8, device=dev)
    outs = []
    for tries in (4, 0):
        ctx = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
        ctx.set_place_tries(tries)
        if not outs:
            ctx.synth_fill(buf.data_ptr(), 0, n)
        for rep in range(2):
            rs = torch.zeros(n // 330 + 2, dtype=torch.int64, device=dev)
            s, c, st = ctx.scan(buf.data_ptr(), n, True, None, rs.data_ptr(), rs.numel())
            assert (st, s.parse_status, s.n_records) == (pkg.OK, pkg.OK, n // 330) and ctx.last_scan_fast()
            outs.append(rs)
        ctx.close()
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    assert torch.equal(outs[0][: n // 330], torch.arange(n // 330, dtype=torch.int64, device=dev) * 330)
    head = buf[: 330 * 4000].cpu().numpy()
    r, off = fqref.offsets(head)
    assert np.array_equal(outs[0][:4000].cpu().numpy().astype(np.uint64), off)


def test_adaptive_line_buffer_is_invisible(fqref, torch, pkg):
    """FQH_OPT_ADAPT_LINES: a context that is given the same big input again tries a second (third ..) allocation for the fast
    path's per-tile lines and keeps the one the input runs faster with.  Whatever it tries and keeps: every call's offsets,
    counts and histograms are the same, and the same as with the option off."""
    dev = torch.device("cuda:0")
    n = (2 << 30) // 330 * 330 + 330 * 7      # just above the 2 GiB from which the library adapts
    d = torch.empty(n + 16, dtype=torch.uint8, device=dev)
    ref = pkg.Ctx(0)
    ref.set_adapt_lines(0)
    ref.synth_fill(d.data_ptr(), 0, n)
    nrec = n // 330
    rs0 = torch.zeros(nrec + 16, dtype=torch.int64, device=dev)
    s0 = ref.scan(d.data_ptr(), n, True, None, rs0.data_ptr(), rs0.numel())[0]
    assert (s0.parse_status, s0.n_records) == (pkg.OK, nrec)
    assert int(rs0[:nrec].sum().item()) == 330 * nrec * (nrec - 1) // 2
    q0 = torch.zeros(150 * 256, dtype=torch.int64, device=dev); b0 = torch.zeros(150 * 8, dtype=torch.int64, device=dev)
    c0 = torch.zeros(8, dtype=torch.int64, device=dev)
    ref.stats(d.data_ptr(), n, 150, q0.data_ptr(), b0.data_ptr(), c0.data_ptr())
    ref.close()
    ctx = pkg.Ctx(0)                           # the option is on by default
    for call in range(7):
        rs = torch.zeros(nrec + 16, dtype=torch.int64, device=dev)
        s = ctx.scan(d.data_ptr(), n, True, None, rs.data_ptr(), rs.numel())[0]
        assert (s.parse_status, s.n_records) == (pkg.OK, nrec) and ctx.last_scan_fast(), call
        assert torch.equal(rs[: nrec + 1], rs0[: nrec + 1]), call
        idx = torch.zeros(1000 * 3, dtype=torch.int64, device=dev)   # the index of the scan just finished is readable whatever buffer it used
        ctx.index_records(idx.data_ptr(), 1000)
        assert int(idx[0].item()) == 0 and int(idx[3].item()) == 330
    for call in range(7):
        q = torch.zeros_like(q0); b = torch.zeros_like(b0); c = torch.zeros_like(c0)
        ctx.stats(d.data_ptr(), n, 150, q.data_ptr(), b.data_ptr(), c.data_ptr())
        assert ctx.last_stats_route() == 1 and torch.equal(q, q0) and torch.equal(b, b0) and torch.equal(c, c0), call
    ctx.close()


def test_own_stream_can_be_made_nonblocking(fqref, torch, pkg):
    """FQH_OPT_OWN_STREAM_NONBLOCKING: a host that overlaps null-stream work with its scans takes the implicit coupling
    out and orders its buffers itself (a synchronize here); results are the same, and the option is refused while a launch is pending."""
    dev = torch.device("cuda:0")
    n = 330 * 20000
    d = torch.empty(n + 16, dtype=torch.uint8, device=dev)
    ctx = pkg.Ctx(0)
    ctx.synth_fill(d.data_ptr(), 0, n)
    ctx.set_own_stream_nonblocking(True)
    rs = torch.zeros(20001, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()                 # (the caller's ordering: the zero-fill above ran on the null stream)
    s = ctx.scan(d.data_ptr(), n, True, None, rs.data_ptr(), rs.numel())[0]
    assert (s.parse_status, s.n_records) == (pkg.OK, 20000)
    assert torch.equal(rs, torch.arange(20001, dtype=torch.int64, device=dev) * 330)
    ctx.scan_launch(d.data_ptr(), n, True, None, rs.data_ptr(), rs.numel())
    with pytest.raises(pkg.FqhError):
        ctx.set_own_stream_nonblocking(False)
    ctx.scan_finish()
    ctx.set_own_stream_nonblocking(False)
    s = ctx.scan(d.data_ptr(), n, True, None, rs.data_ptr(), rs.numel())[0]
    assert s.n_records == 20000
    ctx.close()


def test_adaptive_line_buffer_settles_and_holds_its_memory(torch, pkg):
    """an alternate that measured like the first buffer was given back, the NEXT call took the given-back buffer
    for "a new workspace", lost the first buffer for good and started over — one line buffer of len / 64 bytes leaked every four
    calls, and the input never settled.  Twenty repeated scans of one big input: the context never holds more than 2 + 3 line
    buffers, it settles (fqh_line_buffers: no unsettled input) within the eight calls bench.py warms up with, from then on it
    holds at most two and the device's free memory does not move; fqh_destroy gives everything back."""
    dev = torch.device("cuda:0")
    n = (2 << 30) // 330 * 330 + 330 * 11
    nrec = n // 330
    d = torch.empty(n + 16, dtype=torch.uint8, device=dev)
    rs = torch.zeros(nrec + 16, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    ctx = pkg.Ctx(0)
    ctx.synth_fill(d.data_ptr(), 0, n)
    per_buffer = None
    free_settled = None
    for call in range(20):
        s = ctx.scan(d.data_ptr(), n, True, None, rs.data_ptr(), rs.numel())[0]
        assert (s.parse_status, s.n_records) == (pkg.OK, nrec) and ctx.last_scan_fast(), call
        lb = ctx.line_buffers()
        per_buffer = per_buffer or lb["bytes"] // lb["alive"]
        assert 1 <= lb["alive"] <= 2 + 3 and lb["bytes"] == lb["alive"] * per_buffer, (call, lb)
        if call >= 8:
            assert lb["unsettled"] == 0 and lb["alive"] <= 2, (call, lb)
            free_now = torch.cuda.mem_get_info()[0]
            free_settled = free_settled if free_settled is not None else free_now
            assert free_now == free_settled, (call, free_now, free_settled)
    # a second big input of the same context goes through the same measurements and settles as well
    d2 = torch.empty(n + 16, dtype=torch.uint8, device=dev)
    ctx.synth_fill(d2.data_ptr(), 0, n)
    for call in range(10):
        ctx.scan(d2.data_ptr(), n, True, None, rs.data_ptr(), rs.numel())
        assert ctx.line_buffers()["alive"] <= 2 + 3
    lb = ctx.line_buffers()
    assert lb["unsettled"] == 0 and lb["alive"] <= 2, lb
    ctx.set_adapt_lines(0)
    assert ctx.line_buffers()["unsettled"] == 0
    ctx.close()
    del d2
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    assert torch.cuda.mem_get_info()[0] >= free_before - (64 << 20), (torch.cuda.mem_get_info()[0], free_before)


def test_own_stream_is_ordered_against_the_null_stream(torch, pkg):
    """A caller that never sets a stream works on the legacy null stream (torch's default): the context's own stream must be
    ordered against it.  Here the zero-fill of the offsets array is queued BEHIND a few milliseconds of other null-stream work,
    so a scan on a stream that does not wait for it finishes first and has part of its offsets wiped (seen once the GPU tests
    ran in another order: the own stream was a non-blocking one)."""
    dev = torch.device("cuda:0")
    assert torch.cuda.current_stream().cuda_stream == 0
    n = (64 << 20) // 330 * 330
    nrec = n // 330
    d = torch.empty(n + 16, dtype=torch.uint8, device=dev)
    ctx = pkg.Ctx(0)                      # (no stream given: the context's own)
    try:
        ctx.synth_fill(d.data_ptr(), 0, n)
        ballast = torch.empty(2 << 30, dtype=torch.uint8, device=dev)
        rs = torch.empty(nrec + 8, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        for rep in range(3):
            for _ in range(12):
                ballast.fill_(rep)        # ~ 1 ms each on the null stream
            rs.zero_()                    # ... and the fill the scan must not overtake
            s, c, st = ctx.scan(d.data_ptr(), n, True, None, rs.data_ptr(), nrec + 8)
            assert st == pkg.OK and s.parse_status == pkg.OK and s.n_records == nrec
            got = rs[: nrec + 1].cpu().numpy().astype(np.uint64)    # (a null-stream read: waits for the scan's stream as well)
            assert np.array_equal(got, np.arange(nrec + 1, dtype=np.uint64) * 330), rep
    finally:
        ctx.close()


@pytest.mark.parametrize("lmax", [300, 600])
def test_stats_with_many_rows_over_short_reads(fqref, torch, pkg, lmax):
    """A caller that asks for more than 256 rows and is counted over the exact index gets the record index written by the scan's
    own emit step, sized for reads of that length (one entry per 512 bytes of input).  Short reads overflow it: the index is
    emitted again into an array that is large enough — and which a free + allocation may place at the SAME address, old entries
    and all (the check for "the scan wrote the whole index" used to be made after that replacement: k_stats_long walked whatever
    lay behind the old entries).  And the route: the rows of the single pass follow the READS (a look at the input's first 64 KiB:
    scan_stats_rows), not the caller's arrays, so 600 rows over reads of 100 bases are one pass as well."""
    rng = np.random.default_rng(77)
    gpu = Gpu(torch, pkg.Ctx(0), pkg)   # (a context of its own: the route is pinned)
    data = fuzzgen.valid_file(rng, 60000, maxlen=100, crlf=False)     # ~ 8 MB: 60 000 records, room for ~ 16 000 entries
    r, oq, ob, osc = fqref.stats(data, lmax)
    for single_pass in (False, True):
        gpu.ctx.set_single_pass(single_pass)
        s, gq, gb, gs = gpu.stats(data, lmax)
        assert (s.parse_status, s.n_records) == (r.status, r.n_records) and r.n_records == 60000
        assert np.array_equal(gs, osc) and np.array_equal(gq, oq) and np.array_equal(gb, ob)
        # valid_file's lines hold bytes of every kind, so some batches are counted behind the single pass, or it is given up
        assert gpu.ctx.last_stats_route() in ((0, 2) if single_pass else (0,)), gpu.ctx.last_stats_route()
    # ... and clean reads of 100 bases are counted by that pass alone, under 300 rows and under 600
    clean = b"".join(b"@r%d\n" % i + bytes(rng.choice(fuzzgen.ALPH, 100).tolist()) + b"\n+\n" + bytes(rng.integers(33, 75, 100).astype(np.uint8).tolist()) + b"\n"
                     for i in range(20000))
    r, oq, ob, osc = fqref.stats(clean, lmax)
    gpu.ctx.close()
    gpu = Gpu(torch, pkg.Ctx(0), pkg)   # (the file above has tiles of more than 512 line starts: its context keeps the longer lists — and the exact path)
    s, gq, gb, gs = gpu.stats(clean, lmax)
    assert np.array_equal(gs, osc) and np.array_equal(gq, oq) and np.array_equal(gb, ob)
    assert gpu.ctx.last_stats_route() == 1, gpu.ctx.last_stats_route()
    gpu.ctx.close()
