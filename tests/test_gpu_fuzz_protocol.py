"""Differential fuzz of the HBM-resident sharded protocol (bench.py --gpus N; tools/fuzz_protocol.py runs it for as long as one
likes): random files (valid, damaged, truncated) cut into random byte-range shards — cuts close together and right behind a
newline included — every "rank" a context on ONE stream.  Device recipe first (fqh_shard_prescan_launch /
fqh_shard_rescan_launch / fqh_scan_finish); if every rank's finish says E_AGAIN, the host recipe (fqh_shard_prescan /
fqh_carry_combine / fqh_rescan_launch).  Status, record count and record starts must be the oracle's sequential Parser::each
over the whole file (src/lib.rs:221-304) — the analogue of what parallel_each's gather returns (src/lib.rs:544-564)."""
import time

import numpy as np
import pytest

import fuzzgen

pytestmark = pytest.mark.gpu


def fuzz_protocol(torch, pkg, fqref, seed, budget, max_cases=None):
    """-> (files checked, files that took the host recipe, files with a parse error)"""
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    W = pkg.SHARD_WORDS
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    t_end = time.time() + budget
    cases = agains = errs = 0
    while time.time() < t_end and (max_cases is None or cases < max_cases):
        L = int(rng.choice([20, 75, 150, 300, 3000]))
        data = fuzzgen.valid_file(rng, int(rng.integers(200, 6000) if L < 3000 else rng.integers(20, 200)), maxlen=L,
                                  crlf=bool(rng.random() < 0.15))
        kind = rng.random()
        if kind < 0.2:
            data = fuzzgen.mutate(rng, data, 1)
        elif kind < 0.3:
            data = data[: len(data) - int(rng.integers(1, 300))]
        n = len(data)
        k = int(rng.integers(1, 6))
        cuts = sorted(set(int(x) for x in rng.integers(1, n, k)))
        if rng.random() < 0.3 and cuts:   # two cuts close together
            c0 = cuts[int(rng.integers(0, len(cuts)))]
            cuts = sorted(set(cuts + [min(n - 1, c0 + int(rng.integers(1, 400)))]))
        if rng.random() < 0.3 and cuts:   # a cut right behind a newline
            j = data.find(b"\n", cuts[0])
            if 0 < j + 1 < n:
                cuts = sorted(set(cuts + [j + 1]))
        bounds = [0] + cuts + [n]
        nsh = len(bounds) - 1
        res, idx = fqref.index(data, bufsize=1 << 22)
        all_words = torch.zeros(nsh * W, dtype=torch.int64, device=dev)
        counts = torch.zeros(nsh * 2, dtype=torch.int64, device=dev)
        ctxs, bufs, outs = [], [], []
        for r, (a, b) in enumerate(zip(bounds[:-1], bounds[1:])):
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
        again, starts, total, status, err = 0, [], 0, pkg.OK, None
        for r, c in enumerate(ctxs):
            try:
                s, cout, st = c.scan_finish()
            except pkg.FqhError as e:
                assert e.status == pkg.E_AGAIN, (seed, cases, bounds, e)
                again += 1
                continue
            if status == pkg.OK:
                offs = outs[r].cpu().numpy()
                if not starts:
                    starts.append(int(offs[0]))
                starts += [int(x) for x in offs[1: s.n_records + 1]]
                total += s.n_records
                if s.parse_status != pkg.OK:
                    status, err = s.parse_status, (s.err_record, s.err_offset)
        assert again in (0, nsh), (seed, cases, bounds, again)
        if again:
            agains += 1
            carry, starts, total, status, err = None, [], 0, pkg.OK, None
            for r, c in enumerate(ctxs):
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
                        status, err = s.parse_status, (s.err_record, s.err_offset)
        assert (status, total) == (res.status, res.n_records), (seed, cases, bounds, (status, total), (res.status, res.n_records), again)
        assert starts[: res.n_records] == [int(x) for x in idx[: res.n_records, 0]], (seed, cases, bounds)
        if res.status != pkg.OK:
            errs += 1
        for c in ctxs:
            c.close()
        cases += 1
    torch.cuda.set_stream(torch.cuda.default_stream(dev))
    return cases, agains, errs


@pytest.mark.parametrize("seed", [1, 2])
def test_fuzz_protocol(fqref, seed):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as g
    pkg = g.load_package()
    cases, agains, errs = fuzz_protocol(torch, pkg, fqref, seed, 12.0, max_cases=150)
    assert cases >= 30 and agains >= 5 and errs >= 3, (cases, agains, errs)
