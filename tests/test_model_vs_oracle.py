"""Differential test: whole-buffer model (tests/model.py, the spec of the HIP kernels) against the
streaming oracle (oracle/fqref.c, the restatement of the reference's Parser::each), on random
valid / mutated / garbage inputs — with the real BUFSIZE and with the reference's cfg(fuzzing)
BUFSIZE=64 (src/lib.rs:126-127), where the "record too long" replay is exercised on every input.
"""
import numpy as np
import pytest

import fuzzgen
import model


def check(fqref, data, bufsize):
    res, idx = fqref.index(data, bufsize=bufsize)
    m = model.scan(data, is_final=True, bufsize=bufsize)
    ctx = (bufsize, data)
    assert m["status"] == res.status, ctx
    assert m["n_records"] == res.n_records, ctx
    assert np.array_equal(m["rec_start"][:-1].astype(np.uint64), idx[:, 0]), ctx
    if res.n_records:
        assert int(m["rec_start"][-1]) == res.bytes_consumed, ctx


@pytest.mark.parametrize("seed", range(8))
def test_scan_model_equals_oracle(fqref, seed):
    for tag, data in fuzzgen.corpus(seed, 250):
        check(fqref, data, fqref.BUFSIZE)
        check(fqref, data, 64)


def test_too_long_band_real_bufsize(fqref):
    """Records with lengths in and around [B-15, B] at varying buffer alignments."""
    rng = np.random.default_rng(3)
    B = fqref.BUFSIZE
    for trial in range(40):
        parts = [fuzzgen.valid_record(rng, j) for j in range(int(rng.integers(0, 5)))]
        L = B + int(rng.integers(-20, 3))
        hdr = L - 8
        parts.append(b"@" + b"h" * (hdr - 1) + b"\nA\n+\nB\n")
        parts += [fuzzgen.valid_record(rng, j) for j in range(int(rng.integers(0, 3)))]
        data = b"".join(parts)
        if trial % 4 == 3:
            data = data[: len(data) - int(rng.integers(1, 12))]
        check(fqref, data, B)


@pytest.mark.parametrize("seed", range(3))
def test_stats_model_equals_oracle(fqref, seed):
    for tag, data in fuzzgen.corpus(100 + seed, 120):
        for lmax in (8, 64):
            r, qh, bh, sc = fqref.stats(data, lmax)
            s, mqh, mbh, msc = model.stats(data, lmax, bufsize=fqref.BUFSIZE)
            assert s["status"] == r.status and s["n_records"] == r.n_records
            assert np.array_equal(qh, mqh) and np.array_equal(bh, mbh) and np.array_equal(sc, msc), data


def test_synth_stats(fqref):
    d = fqref.synth(0, 330 * 300)
    r, qh, bh, sc = fqref.stats(d, 150)
    s, mqh, mbh, msc = model.stats(bytes(d), 150)
    assert np.array_equal(qh, mqh) and np.array_equal(bh, mbh) and np.array_equal(sc, msc)
    assert sc[0] == 300 and sc[1] == sc[2] == 300 * 150 and sc[5] == sc[6] == 0
    assert (qh.sum(axis=1) == 300).all() and (bh.sum(axis=1) == 300).all()
