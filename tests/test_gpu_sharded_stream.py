"""configs[4] on one GPU: a file cut into byte-range shards at arbitrary offsets, every shard streamed through its own
pinned ring phase-free (fastq-rs_amd/sharded.py: fqh_shard_align + fqh_stream_*), one exchange, the one-record
stitch at every cut, one sum — must equal the oracle's sequential Parser::each + histogram loop over the whole file
(the reference gathers per-worker results the same way, src/lib.rs:553-559).  The "ranks" run one after the other in
this process; the multi-process form of the same protocol is bench.py --gpus N --stream-gib G
(tests/test_gpu_sharded_stream.py::test_bench_sharded_streamed_three_ranks_one_gpu)."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import fuzzgen

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def env():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as g
    pkg = g.load_package()
    import importlib
    sharded = importlib.import_module("fastq_rs_amd.sharded")
    return torch, pkg, sharded


def run_sharded(env, data, cuts, lmax, slot_bytes=1 << 16):
    """Every "rank" streams its byte range (fqh_shard_stream_run), the exchange is a list, every rank finishes
    (fqh_shard_stream_finish: phase check, stitch, packed first-error key); the sum of the records and the MINIMUM of the keys
    are what the two all-reduces deliver.  -> (status, n_records, histograms, shards)"""
    torch, pkg, sharded = env
    dev = torch.device("cuda:0")
    n = len(data)
    host = (C.c_uint8 * n).from_buffer_copy(data)

    def read_into(addr, off, nbytes):
        C.memmove(addr, C.addressof(host) + off, nbytes)

    bounds = [0] + list(cuts) + [n]
    hist = torch.zeros(8 + lmax * 264, dtype=torch.int64, device=dev)
    sc, qh, bh = hist[:8], hist[8: 8 + lmax * 256], hist[8 + lmax * 256:]
    stats = (lmax, qh.data_ptr(), bh.data_ptr(), sc.data_ptr())
    shards = []
    for r in range(len(bounds) - 1):  # every "rank" has a context (a GPU) of its own; they share the histograms here
        ctx = pkg.Ctx(0)
        shards.append(sharded.stream_shard(ctx, read_into, bounds[r], bounds[r + 1], n, slot_bytes, stats=stats))
        ctx.close()
    # ---- the exchange (lists instead of an all_gather), then every rank's finish; sum and minimum instead of all_reduces
    words = [sh.words() for sh in shards]
    tails = [sh.tail for sh in shards]
    total, key = 0, pkg.NO_ERROR_KEY
    for r, sh in enumerate(shards):
        ctx = pkg.Ctx(0)
        rec, k = sharded.finish(ctx, words, tails, r, sh.head, stats=stats)
        ctx.close()
        total += rec
        key = min(key, k)
    status, err_record = pkg.error_key_unpack(key)
    n_records = total if status == pkg.OK else err_record
    return status, n_records, hist.cpu().numpy().astype(np.uint64), shards


def cut_points(rng, data, k):
    """k cuts: inside lines, right behind a newline, at a record start, and two close together."""
    n = len(data)
    cuts = sorted(set(int(x) for x in rng.integers(n // 10, n - n // 10, k)))
    nl = data.find(b"\n", cuts[0])
    cuts[0] = nl + 1                                  # a line start
    at = data.find(b"\n@", cuts[-1])
    if at > 0:
        cuts[-1] = at + 1                             # (very likely) a record start
    return sorted(set(cuts))


@pytest.mark.parametrize("seed", range(4))
def test_sharded_streamed_equals_whole_file_oracle(env, fqref, seed):
    torch, pkg, sharded = env
    rng = np.random.default_rng(700 + seed)
    lmax = 150
    recs = []
    for i in range(12000):
        L = 150 if seed < 2 else int(rng.integers(1, 151))
        e = b"\r\n" if (seed == 3 and i % 4 == 0) else b"\n"
        seq = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), L, p=[.2475, .2475, .2475, .2475, .01]).tobytes()
        qual = rng.integers(33, 75, L).astype(np.uint8).tobytes()   # '+' and '@' among the quality bytes
        recs.append(b"@r%d/1" % i + e + seq + e + b"+" + e + qual + e)
    data = b"".join(recs)
    cuts = cut_points(rng, data, 3 + seed % 2)
    status, n_records, hist, results = run_sharded(env, data, cuts, lmax, slot_bytes=(1 << 16) if seed % 2 else (1 << 20))
    r, oq, ob, osc = fqref.stats(data, lmax)
    assert (status, n_records) == (r.status, r.n_records) == (pkg.OK, 12000)
    assert np.array_equal(hist[:8], osc), (hist[:8], osc)
    assert np.array_equal(hist[8: 8 + lmax * 256].reshape(lmax, 256), oq)
    assert np.array_equal(hist[8 + lmax * 256:].reshape(lmax, 8), ob)
    # the ranks' pieces tile the file: head + streamed records + tail of every rank
    assert sum(len(x.head) + len(x.tail) for x in results) == sum(
        len(results[i].tail) + len(results[i + 1].head) for i in range(len(results) - 1))
    assert results[0].res.head_len == 0 and results[-1].res.tail_len == 0


@pytest.mark.parametrize("kind", ["truncated", "mismatch", "header", "sep_first_shard"])
def test_sharded_streamed_reports_the_first_error(env, fqref, kind):
    torch, pkg, sharded = env
    rng = np.random.default_rng(31)
    data = bytearray(fuzzgen.valid_file(rng, 6000, maxlen=150, crlf=False))
    if kind == "truncated":
        del data[-7:]
    elif kind == "mismatch":
        k = data.index(b"\n+", len(data) * 2 // 3)
        del data[k - 1]          # one base less in a sequence line of the last shard
    elif kind == "sep_first_shard":
        k = data.index(b"\n+\n", len(data) // 8)
        data[k + 1] = ord("-")   # a separator line without its '+', in the FIRST shard: the later ranks' records must not count
    else:
        k = data.index(b"\n@", len(data) // 2)
        data[k + 1] = ord("x")   # a header that does not start with '@', in a middle shard
    data = bytes(data)
    cuts = [len(data) // 4 + 3, len(data) // 2 - 40, len(data) * 3 // 4 + 11]
    status, n_records, hist, results = run_sharded(env, data, cuts, 150)
    r = fqref.count(data)
    assert r.status != pkg.OK
    assert status != pkg.OK
    if kind != "header":  # (an error in a shard's alignment window is reported at the shard's start: include/fastq_hip.h)
        assert (status, n_records) == (r.status, r.n_records)   # the MINIMUM over the ranks' keys is the error in file order


def test_bench_sharded_streamed_three_ranks_one_gpu():
    """bench.py --gpus 3 --stream-gib G: three processes (gloo, all on cuda:0), real cuts inside records, totals
    checked inside bench.py against the generator's; here: the JSON line is there and says so."""
    env = dict(os.environ, FQH_BENCH_BACKEND="gloo", FQH_BENCH_ONE_GPU="1", MASTER_ADDR="127.0.0.1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
           "--master-port", "29531", os.path.join(ROOT, "bench.py"), "--gpus", "3", "--stream-gib", "0.75", "--slot-mib", "32"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and lines, out.stdout[-2000:] + out.stderr[-2000:]
    j = json.loads(lines[-1])
    assert j["mode"] == "sharded-stream" and j["n_gpus"] == 3 and j["check"]["phases_ok"] and j["check"]["histograms_ok"]
    assert j["records"] == j["check"]["records_expected"]


def test_a_byte_range_too_small_to_settle_its_phase_is_refused(env):
    """A byte range of a few lines cannot tell the line phases apart by itself; the driver then looks at what FOLLOWS it in
    the file (the window only settles the phase and finds the first record start).  Only at the very end of the file is there
    nothing to look at: such a range is settled by its own few lines or refused (FQH_E_ARG), the file is never mis-parsed.  A range without a single line start
    needs no phase (it is stitched across, next test), and an EMPTY range is fine."""
    torch, pkg, sharded = env
    rng = np.random.default_rng(5)
    data = fuzzgen.valid_file(rng, 4000, maxlen=100, seqlen=100, crlf=False)
    r = __import__("oracle.fqref", fromlist=["x"]).count(data)
    k = data.index(b"\n+\n", len(data) // 2)                # the end of a sequence line: [k - 4, k + 8) holds the separator
    status, n_records, hist, shards = run_sharded(env, data, [k - 4, k + 8], 100)   # line and the starts of two more lines:
    assert (status, n_records) == (r.status, r.n_records)                             # settled by the lines behind the range
    for back in range(1, 140, 7):                           # ranges of a few bytes to two lines at the very END of the file,
        try:                                                # where nothing follows: settled (and then right) or refused
            status, n_records, hist, shards = run_sharded(env, data, [len(data) - back], 100)
            assert (status, n_records) == (r.status, r.n_records), back
        except pkg.FqhError as e:
            assert e.status == pkg.E_ARG, back
    status, n_records, hist, shards = run_sharded(env, data, [k + 10, k + 14], 100)   # four bytes inside the quality line
    assert (status, n_records) == (r.status, r.n_records)
    status, n_records, hist, shards = run_sharded(env, data, [k, k], 100)   # (an empty range between two shards)
    assert (status, n_records) == (r.status, r.n_records)


def fuzz_sharded(env, fqref, seed, budget, max_cases=None):
    """Random files (valid, damaged, truncated) cut into random byte-range shards -> (files checked, files with a parse error):
    status and record count must be the oracle's sequential Parser::each over the whole file; histograms too when it is valid."""
    import time
    torch, pkg, sharded = env
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    cases = errs = 0
    while time.time() < t_end and (max_cases is None or cases < max_cases):
        L = int(rng.choice([20, 75, 150, 300]))
        data = fuzzgen.valid_file(rng, int(rng.integers(300, 12000)), maxlen=L, crlf=bool(rng.random() < 0.15))
        kind = rng.random()
        if kind < 0.25:
            data = fuzzgen.mutate(rng, data, 1)
        elif kind < 0.35:
            data = data[: len(data) - int(rng.integers(1, 300))]
        n = len(data)
        k = int(rng.integers(1, 5))
        cuts = sorted(set(int(x) for x in rng.integers(1, n, k)))
        if rng.random() < 0.3:    # two cuts close together: a byte range of a few bytes to a few lines
            c0 = cuts[int(rng.integers(0, len(cuts)))]
            cuts = sorted(set(cuts + [min(n - 1, c0 + int(rng.integers(1, 400)))]))
        if rng.random() < 0.2:    # a cut right behind a newline
            j = data.find(b"\n", cuts[0])
            if 0 < j + 1 < n:
                cuts = sorted(set(cuts + [j + 1]))
        lmax = 150
        try:
            status, n_records, hist, shards = run_sharded(env, data, cuts, lmax, slot_bytes=int(rng.choice([1 << 16, 1 << 18, 1 << 20])))
        except pkg.FqhError as e:
            assert e.status == pkg.E_ARG and min(b - a for a, b in zip(cuts, cuts[1:] + [n])) < pkg.BUFSIZE, (seed, cases, cuts, e)
            continue   # a byte range too small to settle its line phase: refused, not mis-parsed
        r, oq, ob, osc = fqref.stats(data, lmax)
        window_error = any(sh.res.status == pkg.E_HEADER and sh.res.n_records == 0 and sh.lo > 0 for sh in shards)
        if r.status == pkg.OK:
            assert (status, n_records) == (pkg.OK, r.n_records), (seed, cases, cuts, status, n_records, r.n_records)
            assert np.array_equal(hist[:8], osc) and np.array_equal(hist[8: 8 + lmax * 256].reshape(lmax, 256), oq) and np.array_equal(hist[8 + lmax * 256:].reshape(lmax, 8), ob), (seed, cases, cuts)
        else:
            errs += 1
            assert status != pkg.OK, (seed, cases, cuts)
            if not window_error:  # (an error inside a shard's alignment window is reported at the shard's start)
                # the first failing record is the oracle's; so is the kind of the error, unless the record straddles a cut:
                # the stitch holds that record's bytes only up to the next rank's first record start as THAT rank settled it, so
                # a damaged record that swallows a line (its '\n' gone) is incomplete there (TRUNCATED), or shows as the next
                # rank's line phase not fitting the newlines in front of it (HEADER) — where the sequential parser reads on into
                # the next record and names another kind (tools/fuzz_sharded.py seed 341: "+\r" instead of "+\n")
                assert n_records == r.n_records, (seed, cases, cuts, (status, n_records), (r.status, r.n_records))
                assert status == r.status or status in (pkg.E_TRUNCATED, pkg.E_HEADER), (seed, cases, cuts, status, r.status)
        cases += 1
    return cases, errs


@pytest.mark.parametrize("shape", ["inside", "to_boundary", "two_in_a_row", "last_truncated", "last_no_newline"])
def test_byte_ranges_without_a_record_start_are_stitched_across(env, fqref, shape):
    """A byte range that lies inside ONE record holds no record start (FQH_SHARD_PASS): all of its bytes are its tail, and the
    stitch of the next rank with a record start — or the end of the file — is parsed across it.  (tools/fuzz_sharded.py, seed
    311: two cuts sixteen bytes apart gave 'truncated' for a valid file before.)  Against the oracle over the whole file."""
    torch, pkg, sharded = env
    rng = np.random.default_rng(5)
    data = fuzzgen.valid_file(rng, 3000, seqlen=150, crlf=False)
    k = data.index(b"\n@", len(data) // 2) + 1            # a record start in the middle of the file
    rec_end = data.index(b"\n@", k + 1) + 1                # ... and the next one
    if shape == "inside":
        cuts = [k + 40, k + 56]                            # sixteen bytes inside the sequence line
    elif shape == "to_boundary":
        cuts = [k + 40, rec_end]                           # the range ends exactly where the next record begins
    elif shape == "two_in_a_row":
        cuts = [k + 30, k + 50, k + 70]                    # two ranges in a row inside one line of one record
    elif shape == "last_truncated":
        data = data[: k + 100]                             # the file ends inside the sequence line of its last record ...
        cuts = [k - 400, k + 40]                           # ... and the last byte range lies inside that line
    else:
        data = data[: rec_end - 1]                         # the file's last record lacks its '\n' ...
        cuts = [k - 400, rec_end - 50]                     # ... and the last byte range lies inside its quality line
    status, n_records, hist, shards = run_sharded(env, data, cuts, 150)
    r, oq, ob, osc = fqref.stats(data, 150)
    assert (status, n_records) == (r.status, r.n_records), (shape, status, n_records, r.status, r.n_records)
    assert any(sh.res.phase == 0xFFFFFFFE for sh in shards), [sh.res.phase for sh in shards]
    if r.status == pkg.OK:
        assert np.array_equal(hist[:8], osc)
        assert np.array_equal(hist[8: 8 + 150 * 256].reshape(150, 256), oq)
        assert np.array_equal(hist[8 + 150 * 256:].reshape(150, 8), ob)
    else:   # (the oracle's verdict on a last record without its newline is the reference's: src/lib.rs:264-294)
        assert shape in ("last_truncated", "last_no_newline")


@pytest.mark.parametrize("seed", [1, 2])
def test_fuzz_sharded(env, fqref, seed):
    cases, errs = fuzz_sharded(env, fqref, seed, 15.0, max_cases=100)
    assert cases >= 20 and errs >= 1, (cases, errs)
