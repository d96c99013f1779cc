"""Differential fuzz of the ROUTES behind fqh_scan / fqh_stats / fqh_scan_stats: random multi-segment files (read lengths
from 0 to 2500 bp — one length per segment, or ragged by up to 2000 —, id lengths from 1 to 120, CRLF, '+id' lines, damage, truncation, random capacities of the offsets array,
random lmax) against the oracle: offsets, counts, status, maximum record length, histograms.  Drives every loop of
k_emit_fast (one line / two lines / list area / generic), the single pass (k_scan_stats) and its fall-backs.  The idea is the
reference's fuzz targets (fuzz/fuzz_targets/fuzz_target_1.rs:11-18), made differential.  tools/fuzz_routes.py runs the
same function for as long as one likes."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ALPH = np.frombuffer(b"ACGTN", dtype=np.uint8)


def fuzz_routes(torch, pkg, fqref, seed, budget_s, max_cases=None):
    """-> (files checked, scans that kept the fast path, statistics calls that kept the single pass)."""
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)

    def segment(nrec, L, hl, crlf, plus_id, ragged):
        out = []
        e = b"\r\n" if crlf else b"\n"
        for i in range(nrec):
            n = int(rng.integers(max(0, L - ragged), L + 1)) if ragged else L
            h = bytes(rng.integers(48, 123, int(rng.integers(1, hl + 1))).astype(np.uint8).tolist()).replace(b"@", b"a")
            seq = rng.choice(ALPH, n, p=[.2475, .2475, .2475, .2475, .01]).tobytes()
            qual = rng.integers(33, 75, n).astype(np.uint8).tobytes()
            out.append(b"@" + h + e + seq + e + b"+" + (h if plus_id else b"") + e + qual + e)
        return b"".join(out)

    t_end = time.time() + budget_s
    cases = fast = fused = 0
    ctx = None
    while time.time() < t_end and (max_cases is None or cases < max_cases):
        if ctx is None or cases % 4 == 0:
            # (list sizes and back-offs stick to a context — one file of very short lines keeps it off the fast path for good:
            # a fresh one every few files, so that both kinds of history are seen)
            if ctx is not None:
                ctx.close()
            ctx = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
        segs = []
        for _ in range(int(rng.integers(1, 5))):
            L = int(rng.choice([0, 1, 20, 36, 50, 75, 100, 125, 150, 151, 170, 200, 250, 257, 300, 320, 384, 400, 500, 511, 512, 600, 1000, 2500]))
            rec = 2 * L + 12
            segs.append(segment(int(rng.integers(200, 1 + (3 << 20) // max(rec, 40))), L, int(rng.choice([1, 8, 30, 60, 120])),
                                bool(rng.random() < 0.15), bool(rng.random() < 0.2), int(rng.choice([0, 0, 0, 3, 40, 2000]))))
        data = b"".join(segs)
        if rng.random() < 0.25:   # bytes the single pass does not count itself (lower-case bases, qualities above '`'): dumps, in every instance
            b = bytearray(data)
            for p in rng.integers(0, len(b), int(rng.integers(1, 40))):
                if b[p] in b"ACGTN":
                    b[p] |= 0x20
                elif 75 > b[p] >= 48 and b[p - 1] != 10:
                    b[p] = 126
            data = bytes(b)
        if rng.random() < 0.15:   # damage
            b = bytearray(data)
            p = int(rng.integers(0, len(b)))
            b[p] = int(rng.choice(list(b"\n@+xA")))
            data = bytes(b)
        if rng.random() < 0.1:
            data = data[: len(data) - int(rng.integers(1, 300))]
        a = np.frombuffer(data, dtype=np.uint8)
        d = torch.empty(a.size + 16, dtype=torch.uint8, device=dev)
        d[: a.size].copy_(torch.from_numpy(a.copy()))
        res, idx = fqref.index(data)
        starts = idx[:, 0]
        capk = int(rng.integers(0, 3))
        cap = [res.n_records + 5, res.n_records + 1, max(1, res.n_records // 2)][capk]
        rs = torch.full((cap + 4,), -1, dtype=torch.int64, device=dev)
        ctx.set_spec(True)
        ctx.set_bufsize(pkg.BUFSIZE)
        s, c, st = ctx.scan(d.data_ptr(), a.size, True, None, rs.data_ptr(), cap)
        assert (s.parse_status, s.n_records) == (res.status, res.n_records), ("scan", seed, cases)
        assert (st == pkg.E_CAPACITY) == (res.n_records + 1 > cap), ("capacity status", seed, cases)
        got = rs.cpu().numpy()
        k = min(cap, res.n_records)
        assert np.array_equal(got[:k].astype(np.uint64), starts[:k]) and np.all(got[cap:] == -1), ("offsets", seed, cases)
        if res.n_records and capk != 2 and res.status == fqref.OK:
            ends = np.concatenate([starts[1:], [len(data)]]).astype(np.int64)
            assert s.max_record_len == int(np.max(ends - starts.astype(np.int64))), ("maxlen", seed, cases)
        fast += bool(ctx.last_scan_fast())
        lmax = int(rng.choice([36, 50, 64, 96, 100, 128, 150, 152, 170, 192, 200, 256, 300, 320, 384, 401, 500, 511, 512, 600, 1000, 2600]))
        qh = torch.zeros(lmax * 256, dtype=torch.int64, device=dev)
        bh = torch.zeros(lmax * 8, dtype=torch.int64, device=dev)
        sc = torch.zeros(8, dtype=torch.int64, device=dev)
        ctx.set_spec(True)
        if rng.random() < 0.5:
            ctx.set_single_pass(True)   # (forgets the back-off of a single pass that was given up; the other half keeps it)
        if rng.random() < 0.5:
            s2, c2 = ctx.stats(d.data_ptr(), a.size, lmax, qh.data_ptr(), bh.data_ptr(), sc.data_ptr())
        else:
            # (every third of these with an offsets array that is too short: FQH_E_CAPACITY on either route, exact histograms)
            short = res.n_records > 4 and rng.random() < 0.33
            cap2 = res.n_records // 2 if short else res.n_records + 8
            rs2 = torch.full((res.n_records + 8,), -1, dtype=torch.int64, device=dev)
            s2, c2, st2 = ctx.scan_stats(d.data_ptr(), a.size, lmax, qh.data_ptr(), bh.data_ptr(), sc.data_ptr(), d_rec_start=rs2.data_ptr(), cap=cap2)
            assert (st2 == pkg.E_CAPACITY) == short, ("scan_stats capacity status", seed, cases, st2)
            got2 = rs2.cpu().numpy()
            k2 = min(cap2, res.n_records)
            assert np.array_equal(got2[:k2].astype(np.uint64), starts[:k2]) and np.all(got2[max(cap2, k2 + 1):] == -1), ("scan_stats offsets", seed, cases)
        fused += bool(ctx.last_scan_fast())
        r, oq, ob, osc = fqref.stats(a, lmax)
        assert (s2.parse_status, s2.n_records) == (r.status, r.n_records), ("stats", seed, cases)
        assert np.array_equal(sc.cpu().numpy().astype(np.uint64), osc), ("scalars", seed, cases, sc.cpu().numpy(), osc)
        assert np.array_equal(qh.cpu().numpy().astype(np.uint64).reshape(lmax, 256), oq), ("qual", seed, cases)
        assert np.array_equal(bh.cpu().numpy().astype(np.uint64).reshape(lmax, 8), ob), ("base", seed, cases)
        cases += 1
    ctx.close()
    return cases, fast, fused


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_fuzz_routes(fqref, seed):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as g
    pkg = g.load_package()
    cases, fast, fused = fuzz_routes(torch, pkg, fqref, seed, budget_s=25.0, max_cases=40)
    assert cases >= 5, cases
    assert fast >= 1 and fused >= 1, (cases, fast, fused)   # both speculative routes were actually exercised
