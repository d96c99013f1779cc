"""configs[4] on one GPU: a file cut into byte-range shards at arbitrary offsets, every shard streamed through its own
pinned ring phase-free (fastq-rs_amd/sharded.py: fqh_shard_stream_run), one exchange of ten words per rank, the gaps between
the ranks' records parsed under the true line phase (fqh_shard_stream_finish), the sums and one minimum — must equal the oracle's sequential Parser::each + histogram loop over the whole file
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


def run_sharded(env, data, cuts, lmax, slot_bytes=1 << 16, bufsize=None, fail_rank=None, fail_from=None, mapped=None):
    """Every "rank" streams its byte range (fqh_shard_stream_run), the exchange is a list of words, every rank finishes
    (fqh_shard_stream_finish: true-phase check, the parse of the gap in front of it, packed first-error key); the per-rank record
    slots, the sum of the histograms and the MINIMUM of the keys are what the all-reduces deliver, fqh_shard_stream_outcome
    turns them into Parser::each's result.  Every rank has histogram arrays of its own.  fail_rank: that rank's read callback
    raises; fail_from: every rank's read callback raises for bytes at or behind that file offset.
    mapped = the most bytes one map call hands out: the ranks stream their ranges IN PLACE from the (registered) host copy
    (fqh_shard_stream_run_mapped), the read callback only serves windows and gaps.
    -> (status, n_records, histograms, shards)"""
    torch, pkg, sharded = env
    dev = torch.device("cuda:0")
    n = len(data)
    host = (C.c_uint8 * max(1, n)).from_buffer_copy(data if n else b"\0")

    def reader(r):
        def read_into(addr, off, nbytes):
            if r == fail_rank or (fail_from is not None and off + nbytes > fail_from):
                raise OSError("injected")
            assert off + nbytes <= n
            C.memmove(addr, C.addressof(host) + off, nbytes)
        return read_into

    bounds = [0] + list(cuts) + [n]
    world = len(bounds) - 1
    hists = [torch.zeros(8 + lmax * 264, dtype=torch.int64, device=dev) for _ in range(world)]

    def stats_of(r):
        h = hists[r]
        return (lmax, h[8: 8 + lmax * 256].data_ptr(), h[8 + lmax * 256:].data_ptr(), h[:8].data_ptr())

    def ctx_of():
        ctx = pkg.Ctx(0)
        if bufsize is not None:
            ctx.set_bufsize(bufsize)
        return ctx

    shards = []
    for r in range(world):  # every "rank" has a context (a GPU) of its own
        ctx = ctx_of()
        map_at = None
        if mapped:
            ctx.host_register(C.addressof(host), max(1, n))

            def map_at(off, want, r=r):
                if r == fail_rank or (fail_from is not None and off >= fail_from):
                    return None
                assert off < n
                return C.addressof(host) + off, min(n - off, mapped, (fail_from - off) if fail_from is not None else n)
        shards.append(sharded.stream_shard(ctx, reader(r), bounds[r], bounds[r + 1], n, slot_bytes, stats=stats_of(r), map_at=map_at))
        if mapped:
            ctx.host_unregister(C.addressof(host))
        ctx.close()
    # ---- the exchange (a list instead of an all_gather), then every rank's finish; sums and minimum instead of all_reduces
    words = [sh.words() for sh in shards]
    slots, key = [0] * world, pkg.NO_ERROR_KEY
    for r in range(world):
        ctx = ctx_of()
        slots[r], k = sharded.finish(ctx, reader(r), n, words, r, slot_bytes, stats=stats_of(r))
        ctx.close()
        key = min(key, k)
    status, n_records, err_offset = sharded.outcome(slots, key)
    hist = sum(h.cpu().numpy().astype(np.uint64) for h in hists)
    return status, n_records, hist, shards


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
    # every rank parsed by itself under the phase its window settled (no rank deferred), and the pieces tile the file
    assert all(x.res.phase <= 3 and not x.failed for x in results)
    assert results[0].res.head_len == 0 and results[-1].res.tail_len == 0


@pytest.mark.parametrize("kind", ["truncated", "mismatch", "header", "sep_first_shard", "header_at_cut", "swallowed_newline_at_cut",
                                  "crlf_sep_at_cut"])
def test_sharded_streamed_reports_the_first_error(env, fqref, kind):
    """THE error of the sequential parse — kind and record — whatever rank it falls into: inside a rank's records, inside a
    rank's alignment window, in the record that straddles a cut (src/lib.rs:544-547, 561-564)."""
    torch, pkg, sharded = env
    rng = np.random.default_rng(31)
    data = bytearray(fuzzgen.valid_file(rng, 6000, maxlen=150, crlf=False))
    cuts = [len(data) // 4 + 3, len(data) // 2 - 40, len(data) * 3 // 4 + 11]
    if kind == "truncated":
        del data[-7:]
    elif kind == "mismatch":
        k = data.index(b"\n+", len(data) * 2 // 3)
        del data[k - 1]          # one base less in a sequence line of the last shard
    elif kind == "sep_first_shard":
        k = data.index(b"\n+\n", len(data) // 8)
        data[k + 1] = ord("-")   # a separator line without its '+', in the FIRST shard: the later ranks' records must not count
    elif kind == "header":
        k = data.index(b"\n@", len(data) // 2)
        data[k + 1] = ord("x")   # a header that does not start with '@', in a middle shard (inside its alignment window)
    elif kind == "header_at_cut":
        k = data.index(b"\n@", cuts[1])
        data[k + 1] = ord("x")   # ... the first record that starts behind a cut
    elif kind == "swallowed_newline_at_cut":
        k = data.rindex(b"\n", 0, cuts[1])
        del data[k]              # the record that straddles a cut loses a newline in front of the cut: it swallows a line
    else:
        k = data.index(b"\n+\n", cuts[1] - 200)
        data[k + 2: k + 2] = b"\r"   # "+\r\n" is fine; then break the NEXT line's end: '+' 'r' without '\n' ...
        del data[k + 3]              # ... so the separator line runs on into the quality line (seed 341 of the round-3 fuzzer)
    data = bytes(data)
    status, n_records, hist, results = run_sharded(env, data, cuts, 150)
    r = fqref.count(data)
    assert r.status != pkg.OK
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
    assert j["mode"] == "sharded-stream" and j["n_gpus"] == 3 and j["ranks"] == 3
    for name in ("producer", "registered", "pinned_replay"):   # (real producer threads / one pinned block replayed: both sub-runs, same function)
        assert j[name]["check"]["phases_ok"] and j[name]["check"]["histograms_ok"]
        assert j[name]["records"] == j[name]["check"]["records_expected"]


def test_no_byte_range_is_refused(env, fqref):
    """A byte range of a few lines cannot tell the line phases apart by itself; the driver then looks at what FOLLOWS it in
    the file (the window only settles the phase and finds the first record start).  At the very end of the file there is
    nothing to look at: such a range defers (FQH_SHARD_DEFER), and the rank that parses the gap behind the last rank with a
    settled phase reads it under the TRUE phase after the exchange.  Any cut gives the oracle's result and histograms."""
    torch, pkg, sharded = env
    rng = np.random.default_rng(5)
    data = fuzzgen.valid_file(rng, 4000, maxlen=100, seqlen=100, crlf=False)
    r, oq, ob, osc = fqref.stats(data, 100)

    def same(cuts):
        status, n_records, hist, shards = run_sharded(env, data, cuts, 100)
        assert (status, n_records) == (r.status, r.n_records), cuts
        assert np.array_equal(hist[:8], osc) and np.array_equal(hist[8: 8 + 100 * 256].reshape(100, 256), oq), cuts
        assert np.array_equal(hist[8 + 100 * 256:].reshape(100, 8), ob), cuts
        return shards

    k = data.index(b"\n+\n", len(data) // 2)                # the end of a sequence line: [k - 4, k + 8) holds the separator
    same([k - 4, k + 8])                                     # line and the starts of two more lines: settled by the lines behind it
    deferred = 0
    for back in range(1, 140, 7):                            # ranges of a few bytes to two lines at the very END of the file
        shards = same([len(data) - back])
        deferred += shards[-1].res.phase in (pkg.SHARD_DEFER, pkg.SHARD_PASS)
    assert deferred >= 3
    same([k + 10, k + 14])                                   # four bytes inside the quality line
    same([k, k])                                             # an empty range between two shards
    same([len(data) - 300, len(data) - 200, len(data) - 90, len(data) - 40, len(data) - 3])   # five ranks in the last record and a half
    same(sorted(int(x) for x in np.random.default_rng(1).integers(1, 600, 7)))                 # seven cuts in the first two records


def test_ambiguous_files_defer_to_the_exchange(env, fqref):
    """A file whose sequence lines start with '@' and whose quality lines start with '+' parses under two line phases: no
    window settles one (FQH_SHARD_DEFER, formerly refused), the true phase comes with the exchange, and the rank behind the
    last one that could parse by itself reads the rest of the file under it.  Also with a real error far behind the cuts."""
    torch, pkg, sharded = env
    unit = b"@AB\n@CD\n+EF\n+GH\n"
    data = unit * 6000
    n = len(data)
    for cuts in ([n // 4 + 1, n // 2 + 7, 3 * n // 4 + 2], [n // 3, n // 3 + 5, n // 3 + 9], [16 * 100, 16 * 200 + 4]):
        status, n_records, hist, shards = run_sharded(env, data, cuts, 8)
        r, oq, ob, osc = fqref.stats(data, 8)
        assert (status, n_records) == (r.status, r.n_records) == (pkg.OK, 6000), (cuts, status, n_records)
        assert np.array_equal(hist[:8], osc) and np.array_equal(hist[8: 8 + 8 * 256].reshape(8, 256), oq)
        assert any(sh.res.phase == pkg.SHARD_DEFER for sh in shards[1:]), [hex(sh.res.phase) for sh in shards]
    bad = bytearray(data)
    bad[16 * 5000 + 8] = ord("x")      # the separator line of record 5000 loses its '+'
    r = fqref.count(bytes(bad))
    status, n_records, hist, shards = run_sharded(env, bytes(bad), [n // 4 + 1, n // 2 + 7, 3 * n // 4 + 2], 8)
    assert (status, n_records) == (r.status, r.n_records) == (pkg.E_SEP, 5000)


def test_three_lines_on_eight_ranks(env, fqref):
    """Files smaller than the number of ranks, empty files, a file that is one newline: every rank takes part."""
    torch, pkg, sharded = env
    for data in (b"", b"\n", b"@a\nAC\n+\n!!\n", b"@a\nAC\n+\n!!", b"@a\nAC\n+\n!", b"@a\nAC\n", b"x", b"@a\nAC\n+\n!!\n@b\nG\n+\n#\n"):
        r = fqref.count(data)
        n = len(data)
        for cuts in ([min(n, i) for i in range(1, 8)], [n * i // 8 for i in range(1, 8)], [0, 0, n // 2, n // 2, n, n, n]):
            status, n_records, hist, shards = run_sharded(env, data, cuts, 8)
            assert (status, n_records) == (r.status, r.n_records), (data, cuts, status, n_records, r.status, r.n_records)


def test_a_rank_that_fails_still_takes_part(env, fqref):
    """A rank whose read callback fails says so in its words (flags bit 1) and goes on — the others would wait for it forever in
    a collective.  Its bytes are part of the open gap: the LAST rank reads them again in file order, and if its callback can, the
    result is the whole file's; a last rank that cannot read fails with FQH_E_IO, which every rank learns from the minimum —
    unless the sequential parser meets a parse error first."""
    torch, pkg, sharded = env
    rng = np.random.default_rng(9)
    data = bytearray(fuzzgen.valid_file(rng, 3000, maxlen=100, crlf=False))
    cuts = [len(data) // 4, len(data) // 2, len(data) * 3 // 4]
    r = fqref.count(bytes(data))
    status, n_records, hist, shards = run_sharded(env, bytes(data), cuts, 100, fail_rank=2)
    assert (status, n_records) == (pkg.OK, r.n_records) and shards[2].res.flags & 2 and shards[2].res.phase == pkg.SHARD_DEFER
    _, oq, ob, osc = fqref.stats(bytes(data), 100)
    assert np.array_equal(hist[:8], osc) and np.array_equal(hist[8: 8 + 100 * 256].reshape(100, 256), oq)
    assert np.array_equal(hist[8 + 100 * 256:].reshape(100, 8), ob)
    status, n_records, hist, shards = run_sharded(env, bytes(data), cuts, 100, fail_rank=3)
    assert status == pkg.E_IO
    k = data.index(b"\n+\n", len(data) // 8)
    data[k + 1] = ord("-")      # a parse error in rank 0 lies in front of the failure in file order
    r = fqref.count(bytes(data))
    status, n_records, hist, shards = run_sharded(env, bytes(data), cuts, 100, fail_rank=3)
    assert (status, n_records) == (r.status, r.n_records) == (pkg.E_SEP, r.n_records)


def test_unreadable_bytes_behind_a_parse_error_do_not_hide_it(env, fqref):
    """A file whose bytes cannot be read from some offset on: the sequential reader reports a parse error that lies in FRONT of
    them and the I/O failure otherwise, whichever rank's range both fall into (the ring reads ahead of the parser: what is
    already on its way is collected before the failure is reported)."""
    torch, pkg, sharded = env
    rng = np.random.default_rng(10)
    data = bytearray(fuzzgen.valid_file(rng, 12000, maxlen=100, crlf=False))
    n = len(data)
    cuts = [n // 4, n // 2, n * 3 // 4]
    fail_from = n * 11 // 16          # inside rank 2's range, a few slots in
    status, n_records, hist, shards = run_sharded(env, bytes(data), cuts, 100, fail_from=fail_from)
    assert status == pkg.E_IO
    behind = bytearray(data)
    k = behind.index(b"\n+\n", n * 23 // 32)
    behind[k + 1] = ord("-")          # behind the unreadable offset: never seen
    status, n_records, hist, shards = run_sharded(env, bytes(behind), cuts, 100, fail_from=fail_from)
    assert status == pkg.E_IO
    k = data.index(b"\n+\n", n // 2 + n // 32)
    data[k + 1] = ord("-")            # in front of it, in the same rank's range
    r = fqref.count(bytes(data))
    status, n_records, hist, shards = run_sharded(env, bytes(data), cuts, 100, fail_from=fail_from)
    assert (status, n_records) == (r.status, r.n_records) and status == pkg.E_SEP


@pytest.mark.parametrize("bufsize", [256, 69632])
def test_too_long_is_judged_on_file_offsets(env, fqref, bufsize):
    """"Fastq record is too long" (src/lib.rs:278-283) depends on the record's file offset mod 16 (csrc/replay.h): records of
    BUFSIZE - 15 .. BUFSIZE bytes in every rank, and across a cut, are accepted or refused as the sequential parser does."""
    torch, pkg, sharded = env
    rng = np.random.default_rng(bufsize)

    def rec(total):
        body = total - 6
        sl = int(rng.integers(0, body // 2 + 1))
        return b"@" + b"h" * (body - 2 * sl) + b"\n" + b"A" * sl + b"\n+\n" + b"I" * sl + b"\n"

    hits = 0
    for trial in range(6 if bufsize > 1000 else 24):
        parts = []
        for i in range(60):
            x = rng.random()
            L = int(rng.integers(bufsize - 17, bufsize + 1)) if x < 0.08 else int(rng.integers(6, bufsize // 2))
            parts.append(rec(max(6, L)))
        data = b"".join(parts)
        n = len(data)
        cuts = sorted(set(int(x) for x in rng.integers(1, n, 3)))
        big = [i for i, p in enumerate(parts) if len(p) >= bufsize - 17]
        if big:   # one cut inside a record of the band
            at = sum(len(p) for p in parts[: big[len(big) // 2]])
            cuts = sorted(set(cuts + [at + bufsize // 2]))
        r = fqref.count(data, bufsize=bufsize)
        status, n_records, hist, shards = run_sharded(env, data, cuts, 8, slot_bytes=1 << 18, bufsize=bufsize)
        assert (status, n_records) == (r.status, r.n_records), (trial, cuts, status, n_records, r.status, r.n_records)
        hits += r.status == pkg.E_TOO_LONG
    assert hits >= 1


def fuzz_sharded(env, fqref, seed, budget, max_cases=None):
    """Random files (valid, damaged, truncated) cut into random byte-range shards -> (files checked, files with a parse error):
    status and record count must be the oracle's sequential Parser::each over the whole file, for EVERY file and cut; histograms
    too when it is valid.  No byte range is refused."""
    import time
    torch, pkg, sharded = env
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    cases = errs = 0
    while time.time() < t_end and (max_cases is None or cases < max_cases):
        L = int(rng.choice([20, 75, 150, 300, 500]))
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
        lmax = int(rng.choice([150, 150, 64, 300, 512]))
        if rng.random() < 0.15:   # bytes that cannot be read from some offset on: the error the sequential reader meets FIRST
            fail_from = int(rng.integers(1, n))
            slot = 1 << 16
            status, n_records, hist, shards = run_sharded(env, data, cuts, lmax, slot_bytes=slot, fail_from=fail_from)
            r = fqref.count(data)
            # (the reference fills a 68 KiB buffer before it parses it, the ring reads three slots ahead: an error this close in
            # front of the unreadable bytes may be reported either way)
            slack = 2 * 69632 + 3 * slot
            e_off = r.bytes_consumed if r.status != pkg.OK else n   # (the failing record begins where the delivered ones end)
            ctx_ = (seed, cases, cuts, fail_from, (status, n_records), (r.status, r.n_records, e_off))
            if r.status != pkg.OK and e_off + slack < fail_from:
                assert (status, n_records) == (r.status, r.n_records), ctx_
            elif e_off >= fail_from:
                assert status == pkg.E_IO, ctx_
            else:
                assert status == pkg.E_IO or (status, n_records) == (r.status, r.n_records), ctx_
            cases += 1
            continue
        # (a third of the cases streams its ranges IN PLACE from registered memory: fqh_shard_stream_run_mapped, map calls that
        # hand out a slot's worth or less)
        how = int(rng.integers(0, 3))
        status, n_records, hist, shards = run_sharded(env, data, cuts, lmax, slot_bytes=int(rng.choice([1 << 16, 1 << 18, 1 << 20])),
                                                      mapped=[None, None, int(rng.choice([1 << 30, 70001, 4097]))][how])
        r, oq, ob, osc = fqref.stats(data, lmax)
        # status and the number of records delivered before the first error are the sequential parser's, whatever the cuts
        assert (status, n_records) == (r.status, r.n_records), (seed, cases, cuts, (status, n_records), (r.status, r.n_records))
        if r.status == pkg.OK:
            assert np.array_equal(hist[:8], osc) and np.array_equal(hist[8: 8 + lmax * 256].reshape(lmax, 256), oq) and np.array_equal(hist[8 + lmax * 256:].reshape(lmax, 8), ob), (seed, cases, cuts)
        else:
            errs += 1
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
    assert any(sh.res.phase in (pkg.SHARD_PASS, pkg.SHARD_DEFER) for sh in shards), [sh.res.phase for sh in shards]
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


@pytest.mark.parametrize("seed", range(4))
def test_sharded_streamed_in_place_from_registered_memory(env, fqref, seed):
    """fqh_shard_stream_run_mapped: every rank's range is DMA'd from the caller's page-locked memory where it lies (no pinned
    staging slots, no host copy) — same outcome and histograms as the oracle over the whole file; map calls that hand out less
    than a slot (seeds 1, 3), a parse error (seed 2), bytes the map cannot deliver from some offset on (seed 3: the sequential
    reader's FQH_E_IO, or a parse error in front of it)."""
    torch, pkg, sharded = env
    rng = np.random.default_rng(4100 + seed)
    lmax = 150
    data = fuzzgen.valid_file(rng, 9000, maxlen=150)
    if seed == 2:
        data = fuzzgen.mutate(rng, data, 1)
    cuts = cut_points(rng, data, 3)
    fail_from = len(data) * 2 // 3 if seed == 3 else None
    status, n_records, hist, results = run_sharded(env, data, cuts, lmax, slot_bytes=1 << 16, mapped=(1 << 30) if seed % 2 == 0 else 30011,
                                                   fail_from=fail_from)
    if fail_from is None:
        r, oq, ob, osc = fqref.stats(data, lmax)
        assert (status, n_records) == (r.status, r.n_records)
        if r.status == pkg.OK:
            assert np.array_equal(hist[:8], osc)
            assert np.array_equal(hist[8: 8 + lmax * 256].reshape(lmax, 256), oq)
            assert np.array_equal(hist[8 + lmax * 256:].reshape(lmax, 8), ob)
    else:
        # the bytes from fail_from on cannot be had by anybody: the reader's error, as for a failing read callback
        assert status == pkg.E_IO
