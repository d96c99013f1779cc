"""Pins the CPU oracle (oracle/fqref.c) against every golden vector the reference's own tests hold
for this path: the 12 unit tests of src/lib.rs:611-811 and the doc-test of src/lib.rs:474-508
(tests/golden/reference_unit_tests.json, written by tests/golden/make_reference_vectors.py).
"""
import json
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_unit_tests.json")) as f:
    GOLD = json.load(f)


def build_input(parts):
    out = bytearray()
    for p in parts:
        if p[0] == "lit":
            out += p[1].encode("latin-1")
        else:
            out += p[1].encode("latin-1") * p[2]
    return bytes(out)


VECS = {v["name"]: v for v in GOLD["vectors"]}


def test_bufsize_matches_reference(fqref):
    assert GOLD["bufsize"] == fqref.BUFSIZE == 69632


@pytest.mark.parametrize("name", [v["name"] for v in GOLD["vectors"] if v["kind"] == "each"])
def test_each_vectors(fqref, name):
    v = VECS[name]
    data = build_input(v["input"])
    exp = v["expect"]
    res, idx = fqref.index(data)
    assert (res.status == fqref.OK) == exp["ok"], fqref.strerror(res.status)
    if "n_records" in exp:
        assert res.n_records == exp["n_records"]
    for i, r in enumerate(exp.get("records", [])):
        h, s, q = fqref.accessors(data, idx[i])
        assert (h, s, q) == (r["head"].encode(), r["seq"].encode(), r["qual"].encode())
    for i, raw in enumerate(exp.get("raw_write", [])):
        st, qual = int(idx[i][0]), int(idx[i][4])
        assert data[st: st + qual + 1] == raw.encode()
    for i, raw in enumerate(exp.get("owned_write", [])):
        # OwnedRecord::write (records.rs:112-128): '@' head '\n' seq '\n' sep '\n' qual '\n'
        st, head, seq, sep, qual = (int(x) for x in idx[i])
        h, s, q = fqref.accessors(data, idx[i])
        sepline = data[st + seq + 1: st + sep]
        sepline = sepline[:-1] if sepline.endswith(b"\r") else sepline
        assert b"@" + h + b"\n" + s + b"\n" + sepline + b"\n" + q + b"\n" == raw.encode()


def test_error_codes_of_failing_vectors(fqref):
    """The reference tests only pin is_err()/InvalidData; the oracle additionally names the error.
    These follow from the cited lines and are what the GPU path must reproduce."""
    code = lambda n: fqref.count(build_input(VECS[n]["input"])).status
    assert code("missing_lines") == fqref.E_TRUNCATED      # lib.rs:286-291
    assert code("truncated") == fqref.E_TRUNCATED
    assert code("length_mismatch") == fqref.E_LEN           # records.rs:233-238
    assert code("huge_incomplete") == fqref.E_TOO_LONG      # lib.rs:278-283


@pytest.mark.parametrize("name", [v["name"] for v in GOLD["vectors"] if v["kind"] == "record_sets"])
def test_record_sets_vectors(fqref, name):
    v = VECS[name]
    data = build_input(v["input"])
    res, sizes, _ = fqref.record_sets(data)
    assert (res.status == fqref.OK) == v["expect"]["ok"]
    if v["expect"]["ok"]:
        assert int(sizes.sum()) == v["expect"]["total_records"]
        assert sizes[0] == 0  # first RecordSet is always empty (lib.rs:381-391)


@pytest.mark.parametrize("name", [v["name"] for v in GOLD["vectors"] if v["kind"] == "parallel_each"])
def test_parallel_each_vectors(fqref, name):
    v = VECS[name]
    data = build_input(v["input"])
    res, sizes, workers = fqref.record_sets(data, n_threads=v["expect"]["n_threads"])
    assert res.status == fqref.OK
    assert int(workers.sum()) == v["expect"]["sum_of_worker_counts"]
    if "records" in v["expect"]:
        _, idx = fqref.index(data)
        r = v["expect"]["records"][0]
        assert fqref.accessors(data, idx[0])[1] == r["seq"].encode()


def test_bufflen_band(fqref):
    """records <= B-15 always accepted, > B always rejected, at file start exactly B ok."""
    B = fqref.BUFSIZE
    mk = lambda n: b"@" + b"a" * (n - 8) + b"\nA\n+\nB\n"
    assert fqref.count(mk(B)).status == fqref.OK
    assert fqref.count(mk(B + 1)).status == fqref.E_TOO_LONG
    # preceded by a small record the big one no longer starts at buffer offset 0
    small = b"@s\nA\n+\nB\n"
    assert fqref.count(small + mk(B - 15)).status == fqref.OK
    assert fqref.count(small + mk(B + 1)).status == fqref.E_TOO_LONG


def test_unpinned_edge_semantics(fqref):
    """Cases the reference tests do not pin; expectations follow from the cited lines."""
    c = lambda b: (fqref.count(b).status, fqref.count(b).n_records)
    assert c(b"") == (fqref.OK, 0)
    assert c(b"\n") == (fqref.E_HEADER, 0)                       # trailing blank line => header error
    assert c(b"@a\nAC\n+\nII\n\n") == (fqref.E_HEADER, 1)
    assert c(b"@a\n\n+\n\n") == (fqref.OK, 1)                    # empty sequence accepted
    assert c(b"@a\nAC\r\n+\nIII\n") == (fqref.OK, 1)            # raw lengths compared (records.rs:233)
    assert c(b"@a\nAC\n-\nII\n") == (fqref.E_SEP, 0)
    assert c(b"@a\nAC\n") == (fqref.E_TRUNCATED, 0)              # read_sep on empty slice => Incomplete
    assert c(b"@a\nAC\n+") == (fqref.E_TRUNCATED, 0)
    assert c(b"@a\nAC\nx") == (fqref.E_SEP, 0)                   # sep byte checked before newline search
    assert c(b"xa\nAC\n+\nII\n") == (fqref.E_HEADER, 0)
    assert c(b"@a\nAC\n+\n@I\n@b\nGG\n+\n+@\n") == (fqref.OK, 2)  # '@' / '+' as first quality char
    r, idx = fqref.index(b"@a\nAC\r\n+\nIII\n")
    assert fqref.accessors(b"@a\nAC\r\n+\nIII\n", idx[0]) == (b"a", b"AC", b"III")


def test_small_bufsize_equals_large(fqref):
    """cfg(fuzzing) BUFSIZE=64 (lib.rs:126-127) forces refills between almost all records; results
    must not depend on the buffer size while every record fits."""
    rng = np.random.default_rng(7)
    recs = []
    for i in range(200):
        n = int(rng.integers(0, 12))
        seq = bytes(rng.choice(list(b"ACGTN"), n).tolist())
        qual = bytes(rng.integers(33, 74, n).astype(np.uint8).tolist())
        recs.append(b"@r%d\n" % i + seq + b"\n+\n" + qual + b"\n")
    data = b"".join(recs)
    r1, i1 = fqref.index(data, bufsize=64)
    r2, i2 = fqref.index(data)
    r3, i3 = fqref.index(data, bufsize=64, max_read=5)
    assert r1.status == r2.status == r3.status == fqref.OK
    assert r1.n_records == r2.n_records == r3.n_records == 200
    assert np.array_equal(i1, i2) and np.array_equal(i1, i3)


def test_synth_is_valid_fastq_and_counter_based(fqref):
    n = 1000
    d = fqref.synth(0, n * 330)
    res, idx = fqref.index(d)
    assert res.status == fqref.OK and res.n_records == n
    assert np.array_equal(idx[:, 0], np.arange(n, dtype=np.uint64) * 330)
    assert bytes(d[:26]) == b"@SYN.000000000000 1:N:0:1\n"
    # any sub-range can be regenerated independently
    assert np.array_equal(fqref.synth(12345, 1000), d[12345:13345])
    h, s, q = fqref.accessors(d, idx[17])
    assert len(s) == len(q) == 150 and set(s) <= set(b"ACGTN") and min(q) >= 35 and max(q) <= 73
    # quality lines do start with the sentinel characters now and then
    firsts = {bytes(d[int(i[0]) + 179: int(i[0]) + 180]) for i in idx}
    assert b"@" in firsts and b"+" in firsts


def test_kseq_cross_check_counts(fqref, tmp_path):
    """Secondary cross-check with the reference repo's own C comparator (examples/c/parse.c +
    kseq.h, built by oracle/Makefile into oracle/_ref/): record COUNT on well-formed input."""
    exe = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "kseq_count")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/kseq_count not built (reference sources absent)")
    for data in (build_input(VECS["correct"]["input"]), build_input(VECS["windows_lineend"]["input"]),
                 bytes(fqref.synth(0, 330 * 5000))):
        # kseq.h (2011) treats a short read() as EOF, so feed it a regular file, not a pipe
        f = tmp_path / "in.fq"
        f.write_bytes(data)
        with open(f, "rb") as fh:
            out = subprocess.run([exe], stdin=fh, capture_output=True, check=True).stdout
        assert int(out.strip()) == fqref.count(data).n_records
