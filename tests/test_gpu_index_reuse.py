"""The tile index of the last scan describes BYTES, not a pointer: a statistics call after the caller's own kernels have
rewritten the buffer must not count over it (r1 ADVICE / r2 VERDICT: same_scan keyed on pointer identity).  Reuse is now an
explicit statement of the caller (FQH_OPT_REUSE_INDEX); by default every fqh_stats* call reads its input itself, the way the
reference's Parser reads whatever its reader hands it (src/lib.rs:255-303)."""
import numpy as np
import pytest

import fuzzgen

pytestmark = pytest.mark.gpu


def test_stats_after_the_buffer_was_rewritten(fqref):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as g
    pkg = g.load_package()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(8)
    a = fuzzgen.valid_file(rng, 3000, maxlen=100)
    b = fuzzgen.valid_file(rng, 3000, maxlen=100)
    n = min(len(a), len(b))
    a, b = a[:n], b[:n]   # (same length: the second one ends truncated or not, whatever the oracle says)
    d = torch.empty(n + 16, dtype=torch.uint8, device=dev)
    lmax = 100
    for reuse in (False, True):
        ctx = pkg.Ctx(0)
        ctx.set_single_pass(False)          # the route that could reuse an index
        ctx.set_reuse_index(reuse)
        d[:n].copy_(torch.from_numpy(np.frombuffer(a, dtype=np.uint8).copy()))
        s, c, st = ctx.scan(d.data_ptr(), n)
        ra = fqref.count(a)
        assert (s.parse_status, s.n_records) == (ra.status, ra.n_records)
        d[:n].copy_(torch.from_numpy(np.frombuffer(b, dtype=np.uint8).copy()))   # the caller's own write: the library cannot see it
        torch.cuda.synchronize()
        qh = torch.zeros(lmax * 256, dtype=torch.int64, device=dev)
        bh = torch.zeros(lmax * 8, dtype=torch.int64, device=dev)
        sc = torch.zeros(8, dtype=torch.int64, device=dev)
        s2, c2 = ctx.stats(d.data_ptr(), n, lmax, qh.data_ptr(), bh.data_ptr(), sc.data_ptr())
        rb, oq, ob, osc = fqref.stats(b, lmax)
        fresh = (s2.parse_status, s2.n_records) == (rb.status, rb.n_records) and \
            np.array_equal(sc.cpu().numpy().astype(np.uint64), osc) and \
            np.array_equal(qh.cpu().numpy().astype(np.uint64).reshape(lmax, 256), oq)
        if not reuse:
            assert fresh, "default: the statistics call must read the bytes that are there now"
        ctx.close()
