"""The N > 1 host logic driven by REAL kernel output: tests/golden/shard_words_gpu.json holds the words fqh_shard_prescan and
fqh_shard_stream_run / fqh_shard_stream_finish produced on an MI355X for a seeded file cut into 2 and 8 byte ranges (made by
tools/make_shard_fixture.py; the file itself is regenerated here from its seed and checked against the fixture's SHA-256).
CPU: the folds (fqh_carry_combine), the classification of the ranks (fqh_shard_stream_finish for every rank that has no gap to
parse: host arithmetic) and the reduction (fqh_shard_stream_outcome) over those words give the oracle's result — the gather of
Parser::parallel_each and the error it returns, src/lib.rs:544-564."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import fuzzgen
from test_shard_carry import shard_summary, truth_carry

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "shard_words_gpu.json")))


def fixture_file(spec):
    rng = np.random.default_rng(spec["seed"])
    data = bytearray(fuzzgen.valid_file(rng, spec["records"], maxlen=spec["maxlen"], crlf=False))
    for off, byte in spec.get("patch", []):
        data[off] = byte
    return bytes(data)


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g
    p = g.load_package()
    if not os.path.exists(p.LIB_PATH):
        g.build()
    return p


@pytest.mark.parametrize("case", ["hbm_ranks2", "hbm_ranks8", "hbm_ranks8_error_in_rank5"])
def test_prescan_words_of_the_kernels_fold_into_the_true_carries(pkg, fqref, case):
    c = FIX[case]
    data = fixture_file(c["file"])
    assert hashlib.sha256(data).hexdigest() == c["sha256"] and len(data) == c["len"]
    bounds = [0] + c["cuts"] + [len(data)]
    carry = None
    total, first_err = 0, None
    for r, row in enumerate(c["prescan_words"]):
        # what the byte-scan kernel reported for the shard is what the bytes say (numpy) ...
        assert row == list(shard_summary(data, bounds[r], bounds[r + 1])[:3]) + shard_summary(data, bounds[r], bounds[r + 1])[3], (case, r)
        # ... its fold is the parser state at the shard's end ...
        carry = pkg.carry_combine(carry, row[0], row[1], row[2], row[3:7])
        base, nlc, back = truth_carry(data, bounds[r + 1])
        assert (carry.base_offset, carry.nl_count, list(carry.back)) == (base, nlc, back), (case, r)
        # ... and the ranks' emit steps under those carries add up to the oracle's sequential parse
        rs = c["rescan"][r]
        if first_err is None:
            total += rs["n_records"]
            if rs["status"] != pkg.OK:
                first_err = (rs["status"], rs["err_record"])
    o = fqref.count(data)
    assert (o.status, o.n_records) == (c["oracle"]["status"], c["oracle"]["n_records"])
    if o.status == pkg.OK:
        assert total == o.n_records and first_err is None
    else:
        assert first_err == (o.status, o.n_records) and total == o.n_records


@pytest.mark.parametrize("case", ["stream_ranks8", "stream_ranks2_cut_in_record", "stream_ranks2_cut_at_record_start",
                                  "stream_ranks8_error_in_rank5", "stream_ranks8_error_across_cut"])
def test_stream_words_of_the_kernels_reduce_to_the_oracles_result(pkg, fqref, case):
    c = FIX[case]
    data = fixture_file(c["file"])
    assert hashlib.sha256(data).hexdigest() == c["sha256"]
    o = fqref.count(data)
    n = len(c["words"])
    words = np.array(c["words"], dtype=np.uint64)
    bounds = [0] + c["cuts"] + [len(data)]
    a = np.frombuffer(data, dtype=np.uint8)
    slots, keys = [], []
    gaps = 0
    for r in range(n):
        w = c["words"][r]
        assert (w[8], w[9]) == (bounds[r], bounds[r + 1])
        if w[0] == pkg.OK:   # a rank that parsed to its end saw every newline of its range
            assert w[2] == int((a[bounds[r]: bounds[r + 1]] == 10).sum()), (case, r)
        # without a context a rank can only do the host arithmetic; a rank with a gap to parse says so (FQH_E_ARG) — those
        # ranks' outputs come from the GPU run recorded in the fixture
        out = (C.c_uint64 * 2)()
        st = pkg.lib().fqh_shard_stream_finish(None, pkg.READ_FN(), None, len(data), words.ctypes.data, n, r, 1 << 16, 2, 0, None, None, None,
                                               C.byref(out))
        if st == pkg.OK:
            assert [int(out[0]), int(out[1])] == c["finish"][r], (case, r, list(out), c["finish"][r])
        else:
            assert st == pkg.E_ARG
            gaps += 1
        slots.append(c["finish"][r][0])
        keys.append(c["finish"][r][1])
    if "cut_at_record_start" in case:
        assert gaps == 0     # cuts on record boundaries: nothing straddles, every finish is host arithmetic
    elif o.status == pkg.OK:
        assert gaps == n - 1   # every rank but the first parses the record that straddles the cut in front of it
    status, n_records, err_offset = pkg.shard_stream_outcome(min(keys), slots)
    assert (status, n_records) == (o.status, o.n_records) == tuple(c["outcome"][:2])
    if o.status != pkg.OK:
        # the failing record starts where the oracle's delivered records end
        r2, off = fqref.offsets(data)
        end = 0 if o.n_records == 0 else int(off[-1]) + int(np.flatnonzero(a[int(off[-1]):] == 10)[3]) + 1
        assert err_offset == end
