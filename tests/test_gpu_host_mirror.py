"""The C++ mirror of the crate API (fastq-rs_amd/host/fastq.hpp) on the GPU: the reference's own
unit tests ported one-to-one (host_tests), and a differential check of Parser::each, record_sets
(set boundaries!) and parallel_each (per-worker counts) against the oracle on random files."""
import os
import subprocess

import numpy as np
import pytest

import fuzzgen

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "fastq-rs_amd", "host", "bin")


@pytest.fixture(scope="module")
def gpu_ok():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if not os.path.exists(os.path.join(BIN, "host_tests")):
        import __graft_entry__ as g
        g.build()


def test_reference_unit_tests_against_cpp_mirror(gpu_ok):
    out = subprocess.run([os.path.join(BIN, "host_tests")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all host tests passed" in out.stdout
    for t in ("correct", "empty_id", "missing_lines", "truncated", "second_idline", "windows_lineend",
              "length_mismatch", "huge_incomplete", "bufflen", "refset", "refset_incomplete",
              "refset_huge_incomplete", "doctest_parallel_each"):
        assert "ok %s\n" % t in out.stdout


def test_cpp_mirror_under_asan_ubsan_and_tsan(gpu_ok):
    """The same mirrored unit tests, built with AddressSanitizer + UBSan and with ThreadSanitizer (the queues of
    parallel_each and thread_reader, the ring's state machine): no report.  (TSan needs ASLR off on this kernel.)"""
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", TSAN_OPTIONS="ignore_noninstrumented_modules=1 report_signal_unsafe=0")
    for exe, pre in (("host_tests_asan", []), ("host_tests_tsan", ["setarch", "x86_64", "-R"])):
        out = subprocess.run(pre + [os.path.join(BIN, exe)], capture_output=True, text=True, timeout=900, env=env)
        assert out.returncode == 0, exe + out.stdout[-2000:] + out.stderr[-4000:]
        assert "all host tests passed" in out.stdout
        assert "Sanitizer" not in out.stderr, out.stderr[-4000:]


def test_leases_queues_and_the_filler_thread_under_tsan_and_asan(gpu_ok, tmp_path):
    """Round 6's threads — RecordSets that give their ring slot back from whatever thread drops them, parallel_each's queues that
    poll before they sleep, the parser's own filler thread (Options::read_ahead) — through each / record_sets / parallel_each(3)
    of the sanitizer builds, on slots small enough that sets straddle them and leases pile up: no report, same output as the
    plain build."""
    rng = np.random.default_rng(77)
    path = tmp_path / "san.fq"
    path.write_bytes(fuzzgen.valid_file(rng, 6000, maxlen=150))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", TSAN_OPTIONS="ignore_noninstrumented_modules=1 report_signal_unsafe=0")
    want = subprocess.run([os.path.join(BIN, "host_tests"), "--dump", str(path), "3", "69632", "65536"], capture_output=True, text=True, timeout=600)
    assert want.returncode == 0
    for exe, pre in (("host_tests_asan", []), ("host_tests_tsan", ["setarch", "x86_64", "-R"])):
        out = subprocess.run(pre + [os.path.join(BIN, exe), "--dump", str(path), "3", "69632", "65536"], capture_output=True, text=True,
                             timeout=900, env=env)
        assert out.returncode == 0, exe + out.stdout[-2000:] + out.stderr[-4000:]
        assert "Sanitizer" not in out.stderr, out.stderr[-4000:]
        assert out.stdout == want.stdout


SETS_MSG = {4: "Truncated input file.", 5: "Fastq record is too long."}


def run_dump(path, threads, bufsize, slot):
    out = subprocess.run([os.path.join(BIN, "host_tests"), "--dump", path, str(threads), str(bufsize), str(slot)],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = {l.split(" ", 1)[0]: l.split(" ", 1)[1] for l in out.stdout.strip().split("\n")}
    # Options::read_ahead (a filler thread of the parser's own; the reader fills every read): everything the calling thread's
    # parser delivers, to the byte
    for k in ("each", "eachsum", "sets", "setsum", "workers", "worksum"):
        assert lines["ahead_" + k] == lines[k], (k, lines[k], lines["ahead_" + k], bufsize, slot)
    return lines


@pytest.mark.parametrize("seed", range(6))
def test_each_sets_parallel_each_equal_oracle(gpu_ok, fqref, tmp_path, seed):
    rng = np.random.default_rng(500 + seed)
    if seed < 3:
        data = fuzzgen.valid_file(rng, 4000, maxlen=150)
    elif seed == 3:
        data = fuzzgen.mutate(rng, fuzzgen.valid_file(rng, 2000, maxlen=150), 1)
    elif seed == 4:
        B = fqref.BUFSIZE
        data = fuzzgen.valid_file(rng, 300) + b"@" + b"h" * (B - 12) + b"\nA\n+\nB\n" + fuzzgen.valid_file(rng, 300)
    else:
        data = fuzzgen.valid_file(rng, 1500, maxlen=150)[:-7]
    path = tmp_path / "in.fq"
    path.write_bytes(data)
    threads = 3
    for bufsize, slot in ((fqref.BUFSIZE, 1 << 20), (fqref.BUFSIZE, 1 << 16), (64, 1 << 16)):
        if bufsize == 64 and seed != 3:
            data2 = fuzzgen.valid_file(rng, 200, maxlen=8)
            path.write_bytes(data2)
        else:
            data2 = data
        got = run_dump(str(path), threads, bufsize, slot)
        r = fqref.count(data2, bufsize=bufsize)
        _, idx = fqref.index(data2, bufsize=bufsize)
        bases = sum(len(fqref.accessors(data2, row)[1]) for row in idx)
        each = got["each"].split(" ", 2)
        assert (int(each[0]), int(each[1])) == (r.n_records, bases)
        assert each[2] == (fqref.strerror(r.status) if r.status else "ok")
        # a pipe that hands out 3001 bytes per read(), Options::low_latency: the records and the verdict of the reference under THAT
        # reader (the "too long" band depends on the read sizes, src/buffer.rs:74-100: the mirror replays them, csrc/replay.h)
        rp = fqref.count(data2, bufsize=bufsize, max_read=3001)
        pipe = got["pipe"].split(" ", 2)
        assert (int(pipe[0]), pipe[2]) == (rp.n_records, fqref.strerror(rp.status) if rp.status else "ok"), (bufsize, slot, got["pipe"])
        rs, sizes, workers = fqref.record_sets(data2, n_threads=threads, bufsize=bufsize)
        s_sizes, s_err = got["sets"].rsplit(" ", 1) if got["sets"].endswith("ok") else got["sets"].split(" ", 1)
        want_sizes = ",".join(str(int(x)) for x in sizes) + ","
        assert s_sizes == want_sizes, (bufsize, slot)
        want_err = "ok" if rs.status == 0 else SETS_MSG.get(rs.status, fqref.strerror(rs.status))
        assert s_err == want_err
        w_counts, w_err = got["workers"].rsplit(" ", 1) if got["workers"].endswith("ok") else got["workers"].split(" ", 1)
        assert w_err == want_err
        if rs.status == 0:
            assert w_counts == ",".join(str(int(x)) for x in workers) + ","
        # what the consumers SAW: the records of each(), of the sets (views into ring slots, kept 40 sets deep) and of the workers
        want_sum = _digest_sum(fqref, data2, idx)
        assert int(got["eachsum"]) == want_sum
        n_in_sets = int(sum(sizes))   # (the two modes may draw the "too long" line at different records: index without it)
        _, idx_all = fqref.index(data2, bufsize=1 << 24)
        assert got["setsum"].split() == [str(n_in_sets), str(_digest_sum(fqref, data2, idx_all[:n_in_sets]))], (bufsize, slot)
        if rs.status == 0:
            assert int(got["worksum"]) == _digest_sum(fqref, data2, idx_all[:n_in_sets])


def _digest_sum(fqref, data, idx):
    """host_tests.cpp: digest() summed over the records — FNV-1a over head | 0xFF | seq | 0xFF | qual | 0xFF | raw | 0xFF."""
    M = (1 << 64) - 1
    total = 0
    for row in idx:
        head, seq, qual = fqref.accessors(data, row)
        start = int(row[0])
        raw = data[start: start + int(row[4]) + 1]
        h = 1469598103934665603
        for part in (head, seq, qual, raw):
            for x in part:
                h = ((h ^ x) * 1099511628211) & M
            h = ((h ^ 0xFF) * 1099511628211) & M
        total = (total + h) & M
    return total


@pytest.mark.parametrize("seed", range(3))
def test_arbitrary_bytes_through_parallel_each(gpu_ok, fqref, tmp_path, seed):
    """The reference's second fuzz target (fuzz/fuzz_targets/fuzz_target_2.rs:10-22: arbitrary bytes through parallel_each(3, ..),
    never a panic), differential: garbage, files with several mutations and truncated files through each, record_sets and
    parallel_each(3) of the C++ mirror — status, counts, set sizes, per-worker counts and the records' bytes are the oracle's,
    at BUFSIZE 64 (cfg(fuzzing), src/lib.rs:126-127) and 68 KiB."""
    rng = np.random.default_rng(6100 + seed)
    path = tmp_path / "fz.fq"
    cases = []
    for i in range(10):
        k = i % 5
        if k == 0:
            cases.append(fuzzgen.garbage(rng, int(rng.integers(0, 3000))))
        elif k == 1:
            cases.append(fuzzgen.mutate(rng, fuzzgen.valid_file(rng, int(rng.integers(1, 400)), maxlen=30), int(rng.integers(2, 6))))
        elif k == 2:
            d = fuzzgen.valid_file(rng, int(rng.integers(1, 300)), maxlen=20)
            cases.append(d[: int(rng.integers(0, len(d) + 1))])
        elif k == 3:
            cases.append(fuzzgen.valid_file(rng, 200, maxlen=12) + fuzzgen.garbage(rng, 100) + fuzzgen.valid_file(rng, 50, maxlen=12))
        else:
            cases.append(bytes(rng.integers(0, 256, int(rng.integers(1, 2000))).astype(np.uint8).tolist()))
    for data in cases:
        path.write_bytes(data)
        for bufsize, slot in ((64, 4096), (fqref.BUFSIZE, 1 << 16)):
            got = run_dump(str(path), 3, bufsize, slot)
            r = fqref.count(data, bufsize=bufsize)
            _, idx = fqref.index(data, bufsize=bufsize)
            each = got["each"].split(" ", 2)
            assert (int(each[0]), each[2]) == (r.n_records, fqref.strerror(r.status) if r.status else "ok"), (data, bufsize)
            assert int(got["eachsum"]) == _digest_sum(fqref, data, idx)
            rs, sizes, workers = fqref.record_sets(data, n_threads=3, bufsize=bufsize)
            s_sizes, s_err = got["sets"].rsplit(" ", 1) if got["sets"].endswith("ok") else got["sets"].split(" ", 1)
            want_err = "ok" if rs.status == 0 else SETS_MSG.get(rs.status, fqref.strerror(rs.status))
            assert (s_sizes, s_err) == (",".join(str(int(x)) for x in sizes) + ",", want_err), (data, bufsize)
            n_in_sets = int(sum(sizes))
            _, idx_all = fqref.index(data, bufsize=1 << 24)
            assert got["setsum"].split() == [str(n_in_sets), str(_digest_sum(fqref, data, idx_all[:n_in_sets]))]
            w_counts, w_err = got["workers"].rsplit(" ", 1) if got["workers"].endswith("ok") else got["workers"].split(" ", 1)
            assert w_err == want_err
            if rs.status == 0:
                assert w_counts == ",".join(str(int(x)) for x in workers) + ","
                assert int(got["worksum"]) == _digest_sum(fqref, data, idx_all[:n_in_sets])


def _sized_record(rng, total):
    """A valid record of exactly `total` bytes (>= 6): '@' h '\\n' s '\\n+\\n' q '\\n' with |s| == |q|."""
    body = total - 6
    s = int(rng.integers(0, body // 2 + 1))
    h = body - 2 * s
    return b"@" + b"h" * h + b"\n" + b"A" * s + b"\n+\n" + b"I" * s + b"\n"


@pytest.mark.parametrize("seed", range(4))
def test_pipes_short_reads_and_the_first_record(gpu_ok, fqref, tmp_path, seed):
    """A reader that comes back with at most CAP bytes per read() (the oracle's max_read; src/buffer.rs:74-100 takes what one
    read() returns): whether a record of BUFSIZE - 15 .. BUFSIZE bytes is "too long" then depends on the sizes of the reads.  The
    mirror notes its own reads and replays the reference's (csrc/replay.h): Parser::each's records and error and record_sets'
    cuts equal the oracle's under the same reader, for caps from 1 to 4096 bytes and for a reader that fills every read.  And
    record 0 reaches the closure after no more input than the reference needs to hold it (src/lib.rs:264-275), not after a
    whole ring slot."""
    rng = np.random.default_rng(4200 + seed)
    B = 256 if seed < 3 else fqref.BUFSIZE
    sizes = []
    for _ in range(60 if B == 256 else 6):
        r = int(rng.integers(0, 10))
        sizes.append(int(rng.integers(6, B // 2)) if r < 6 else int(rng.integers(B - 20, B + 1)))
    if seed == 1:
        sizes.append(B + 1 + int(rng.integers(0, 40)))          # certainly too long, behind records the band decides
    data = b"".join(_sized_record(rng, t) for t in sizes)
    if seed == 2:
        data = data[:-3]                                        # a truncated tail
    path = tmp_path / "pipe.fq"
    path.write_bytes(data)
    slot = 1 << 16 if B == 256 else 1 << 20
    verdicts = set()
    for cap in (0, 1, 5, 16, 37, 100, 255, 4095, 4096):
        out = subprocess.run([os.path.join(BIN, "host_tests"), "--pipe", str(path), str(cap), str(B), str(slot)],
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout + out.stderr
        got = {l.split(" ", 1)[0]: l.split(" ", 1)[1] for l in out.stdout.strip().split("\n")}
        r = fqref.count(data, bufsize=B, max_read=cap)
        each = got["each"].split(" ", 2)
        assert (int(each[0]), each[2]) == (r.n_records, fqref.strerror(r.status) if r.status else "ok"), (cap, got["each"], r.status, r.n_records)
        verdicts.add((r.status, r.n_records))
        rs, set_sizes, _ = fqref.record_sets(data, n_threads=1, bufsize=B, max_read=cap)
        s_sizes, s_err = got["sets"].rsplit(" ", 1) if got["sets"].endswith("ok") else got["sets"].split(" ", 1)
        want_sizes = ",".join(str(int(x)) for x in set_sizes) + ","
        assert s_sizes == want_sizes and s_err == ("ok" if rs.status == 0 else SETS_MSG.get(rs.status, fqref.strerror(rs.status))), (cap, B)
        if r.n_records:
            assert int(got["first"]) <= max(sizes[0] + (cap or fqref.BUFSIZE), min(fqref.BUFSIZE, len(data))), (cap, got["first"])
    if B == 256:
        assert len(verdicts) > 1, verdicts      # (the reader's read sizes DID decide some record of the band)


@pytest.mark.parametrize("seed", range(5))
def test_each_zipped_equals_oracle(gpu_ok, fqref, tmp_path, seed):
    """each_zipped (src/lib.rs:577-609): two files of different lengths, a scripted callback with random advance flags
    (single reads interleaved in paired data are skipped that way); every callback's pair of records, the returned
    (bool, bool) and the error of a bad file equal the oracle's restatement."""
    rng = np.random.default_rng(900 + seed)
    n1, n2 = int(rng.integers(50, 400)), int(rng.integers(50, 400))
    mk = lambda n, tag: b"".join(fuzzgen.valid_record(rng, i, maxlen=60).replace(b"@r", b"@" + tag, 1) for i in range(n))
    d1, d2 = mk(n1, b"a"), mk(n2, b"b")
    if seed == 3:
        d2 = d2[:-9]                      # truncated second file: the error surfaces when its iterator gets there
    if seed == 4:
        d1 = fuzzgen.mutate(rng, d1, 1)
    flags = "".join(str(int(x)) for x in rng.choice([1, 2, 3, 3, 3], int(rng.integers(1, 40)))) if seed else "3"
    if seed == 2:
        flags += "0"                      # the callback stops the walk
    (tmp_path / "a.fq").write_bytes(d1)
    (tmp_path / "b.fq").write_bytes(d2)
    for bufsize, slot in ((fqref.BUFSIZE, 1 << 16), (fqref.BUFSIZE, 1 << 20)):
        out = subprocess.run([os.path.join(BIN, "host_tests"), "--zip", str(tmp_path / "a.fq"), str(tmp_path / "b.fq"), flags,
                              str(bufsize), str(slot)], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout + out.stderr
        lines = out.stdout.strip().split("\n")
        st, fin, trace = fqref.each_zipped(d1, d2, bytes(int(c) for c in flags), bufsize=bufsize)
        _, i1 = fqref.index(d1, bufsize=bufsize)
        _, i2 = fqref.index(d2, bufsize=bufsize)
        NONE = np.uint64(0xFFFFFFFFFFFFFFFF)
        want = []
        for a, b in trace:
            h1 = "-" if a == NONE else fqref.accessors(d1, i1[int(a)])[0].decode("latin-1")
            h2 = "-" if b == NONE else fqref.accessors(d2, i2[int(b)])[0].decode("latin-1")
            want.append("call %s %s" % (h1, h2))
        assert lines[:-1] == want
        z = lines[-1].split(" ", 3)
        if st == 0:
            assert z == ["zipped", str(int(fin[0])), str(int(fin[1])), "ok"]
        else:
            assert z[3] == fqref.strerror(st)


def test_fastq_count_cli(gpu_ok, fqref, tmp_path):
    """examples/fastq-count.rs behaviour: decimal count on stdout; failure on an invalid file."""
    d = bytes(fqref.synth(0, 330 * 20000))
    p = tmp_path / "a.fq"
    p.write_bytes(d)
    exe = os.path.join(BIN, "fastq_count")
    assert subprocess.run([exe, str(p)], capture_output=True, text=True).stdout.strip() == "20000"
    assert subprocess.run([exe, "--threads", "2", str(p)], capture_output=True, text=True).stdout.strip() == "20000"
    with open(p, "rb") as fh:
        assert subprocess.run([exe, "-"], stdin=fh, capture_output=True, text=True).stdout.strip() == "20000"
    p.write_bytes(d[:-5])
    bad = subprocess.run([exe, str(p)], capture_output=True, text=True)
    assert bad.returncode != 0 and "Possibly truncated input file" in bad.stderr


def test_fastq_count_on_gzip_input(gpu_ok, fqref, tmp_path):
    """parse_path (src/lib.rs:167-196): gzip input is sniffed, decoded on a thread_reader thread and
    scanned on the GPU; the count equals the oracle's on the plain bytes, single- and multi-threaded."""
    import gzip
    rng = np.random.default_rng(4242)
    data = fuzzgen.valid_file(rng, 20000, maxlen=150)
    want = fqref.count(data).n_records
    cut = len(data) // 2
    (tmp_path / "in.fq").write_bytes(data)
    (tmp_path / "in.fq.gz").write_bytes(gzip.compress(data[:cut]) + gzip.compress(data[cut:]))
    names = ["in.fq", "in.fq.gz"]
    try:  # lz4 as well (src/lib.rs:137-141 names it), when the box has a liblz4 to write the test frame with
        from test_host_parse_path import lz4_frame
        (tmp_path / "in.fq.lz4").write_bytes(lz4_frame(data[:cut]) + lz4_frame(data[cut:], block_linked=True, block_size_id=4))
        names.append("in.fq.lz4")
    except OSError:
        pass
    import bz2, lzma  # the other formats niffler sniffs: decoded by the system's libraries, bound at run time
    (tmp_path / "in.fq.bz2").write_bytes(bz2.compress(data))
    (tmp_path / "in.fq.xz").write_bytes(lzma.compress(data[:cut], format=lzma.FORMAT_XZ) + lzma.compress(data[cut:], format=lzma.FORMAT_XZ))
    names += ["in.fq.bz2", "in.fq.xz"]
    try:
        from test_host_parse_path import zstd_frame
        (tmp_path / "in.fq.zst").write_bytes(zstd_frame(data))
        names.append("in.fq.zst")
    except OSError:
        pass
    for name in names:
        for extra in ([], ["--threads", "3"]):
            out = subprocess.run([os.path.join(BIN, "fastq_count"), str(tmp_path / name)] + extra,
                                 capture_output=True, text=True, timeout=600)
            assert out.returncode == 0, out.stdout + out.stderr
            assert int(out.stdout.strip()) == want == 20000
    bad = data[:cut] + b"\n" + data[cut:]
    (tmp_path / "bad.fq.gz").write_bytes(gzip.compress(bad))
    out = subprocess.run([os.path.join(BIN, "fastq_count"), str(tmp_path / "bad.fq.gz")],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 101 and fqref.strerror(fqref.count(bad).status) in out.stderr


def test_each_sharded_cpp_mirror_reports_a_failing_reader(gpu_ok, tmp_path):
    """A read callback that throws inside fastq::each_sharded: the rank still takes part in the exchange and the reductions (it
    would otherwise leave the other ranks waiting in a collective) and throws afterwards — an I/O failure, not a parse
    error; with a parse error in front of the failing bytes, the parse error wins (file order)."""
    rng = np.random.default_rng(92)
    data = bytearray(fuzzgen.valid_file(rng, 9000, maxlen=120, crlf=False))
    path = tmp_path / "f.fastq"
    path.write_bytes(bytes(data))
    out = subprocess.run([os.path.join(BIN, "host_tests"), "--sharded", str(path), "120", "fail-io"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.startswith("err ") and "read callback failed" in out.stdout, out.stdout + out.stderr
    k = data.index(b"\n+", len(data) // 8)
    data[k + 1] = ord("-")
    path.write_bytes(bytes(data))
    out = subprocess.run([os.path.join(BIN, "host_tests"), "--sharded", str(path), "120", "fail-io"], capture_output=True, text=True, timeout=600)
    assert out.stdout.strip() == "err Sequence and quality not separated by +", out.stdout + out.stderr


@pytest.mark.parametrize("kind", ["valid", "crlf", "mismatch"])
def test_each_sharded_cpp_mirror_equals_oracle(gpu_ok, fqref, tmp_path, kind):
    """fastq::each_sharded (host/fastq.hpp) over fqh_shard_stream_run / fqh_shard_stream_finish as the only rank: record count,
    totals and histogram checksums equal the oracle's Parser::each + histogram loop; a parse error is thrown with the
    reference's message (src/lib.rs:561-564: parallel_each returns the parse error)."""
    rng = np.random.default_rng(91)
    data = bytearray(fuzzgen.valid_file(rng, 9000, maxlen=120, crlf=(kind == "crlf")))
    if kind == "mismatch":
        k = data.index(b"\n+", len(data) // 2)
        del data[k - 1]
    path = tmp_path / "f.fastq"
    path.write_bytes(bytes(data))
    lmax = 120
    out = subprocess.run([os.path.join(BIN, "host_tests"), "--sharded", str(path), str(lmax)], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    r, oq, ob, osc = fqref.stats(np.frombuffer(bytes(data), dtype=np.uint8), lmax)
    if kind == "mismatch":
        assert out.stdout.strip() == "err Sequence and quality length mismatch" and r.status == 3
        return
    sq = int((oq.reshape(-1).astype(object) * [(i % 251 + 1) for i in range(lmax * 256)]).sum()) % (1 << 64)
    sb = int((ob.reshape(-1).astype(object) * [(i % 13 + 1) for i in range(lmax * 8)]).sum()) % (1 << 64)
    assert out.stdout.split() == ["ok", str(r.n_records), str(int(osc[0])), str(int(osc[1])), str(sq), str(sb)], out.stdout
