"""scan -> select -> gather -> write on the GPU (fqh_record_flags, fqh_gather_records) against the CPU
statement of the same pipeline over the oracle's records: flags = Record::validate_dna /
validate_dnan (src/records.rs:19-33) on seq(), output = the selected records' raw bytes back to back
(what RefRecord::write, src/records.rs:93-96, emits)."""
import numpy as np
import pytest

import fuzzgen

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as g
    return torch, g.load_package()


def cpu_pipeline(fqref, data):
    res, idx = fqref.index(data)
    flags, raws = [], []
    for row in idx:
        _, seq, _ = fqref.accessors(data, row)
        dna = all(c in b"ACGT" for c in seq)
        dnan = all(c in b"ACGTN" for c in seq)
        flags.append((1 if dna else 0) | (2 if dnan else 0))
        start, qual = int(row[0]), int(row[4])
        raws.append(data[start: start + qual + 1])
    return res, np.array(flags, dtype=np.uint8), raws


@pytest.mark.parametrize("seed", range(4))
def test_flags_and_gather_equal_cpu_pipeline(fqref, env, seed):
    torch, pkg = env
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(4100 + seed)
    # fuzzgen: 10 % of the records have arbitrary printable sequence bytes, the rest ACGTN with N's
    data = fuzzgen.valid_file(rng, 5000 if seed else 40000, maxlen=[120, 300, 17, 150][seed], crlf=(seed == 1))
    res, flags, raws = cpu_pipeline(fqref, data)
    n = res.n_records
    assert 0 < int((flags == 3).sum()) < n and int((flags & 2 == 0).sum()) > 0
    d = torch.empty(len(data) + 16, dtype=torch.uint8, device=dev)
    d[: len(data)].copy_(torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()))
    ctx = pkg.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    rs = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    s, c, st = ctx.scan(d.data_ptr(), len(data), True, None, rs.data_ptr(), n + 1)
    assert s.n_records == n
    idx = torch.zeros(n * 24, dtype=torch.uint8, device=dev)
    ctx.index_records(idx.data_ptr(), n)
    gflags = torch.full((n,), 0xFF, dtype=torch.uint8, device=dev)
    ctx.record_flags(d.data_ptr(), len(data), idx.data_ptr(), n, gflags.data_ptr())
    assert np.array_equal(gflags.cpu().numpy(), flags)
    for mask, want in ((3, 3), (2, 2), (2, 0), (0, 0), (3, 2)):
        sel = [i for i in range(n) if (int(flags[i]) & mask) == want]
        expect = b"".join(raws[i] for i in sel)
        st, ns, nb = ctx.gather_records(d.data_ptr(), len(data), idx.data_ptr(), n, gflags.data_ptr(), mask, want, None, 0)
        assert (st, ns, nb) == (pkg.OK, len(sel), len(expect))                       # sizing call
        out = torch.zeros(nb + 16, dtype=torch.uint8, device=dev)
        st, ns, nb2 = ctx.gather_records(d.data_ptr(), len(data), idx.data_ptr(), n, gflags.data_ptr(), mask, want,
                                         out.data_ptr(), nb)
        assert (st, ns, nb2) == (pkg.OK, len(sel), nb)
        assert out[:nb].cpu().numpy().tobytes() == expect
        assert int(out[nb:].sum()) == 0                                              # nothing past the end
        if nb > 100:                                                                 # too small: reported, no overrun
            small = torch.zeros(nb, dtype=torch.uint8, device=dev)
            st, _, nb3 = ctx.gather_records(d.data_ptr(), len(data), idx.data_ptr(), n, gflags.data_ptr(), mask, want,
                                            small.data_ptr(), nb - 50)
            assert st == pkg.E_CAPACITY and nb3 == nb
            got = small.cpu().numpy()
            assert int(got[nb - 50:].sum()) == 0
            fit, pos = 0, 0                       # records that fit completely are written, the rest are not
            for i in sel:
                if pos + len(raws[i]) <= nb - 50:
                    fit = pos + len(raws[i])
                pos += len(raws[i])
            assert fit > 0 and got[:fit].tobytes() == expect[:fit]
    # the filtered output is itself a valid FASTQ file with exactly the selected records
    sel = [i for i in range(n) if int(flags[i]) == 3]
    assert fqref.count(b"".join(raws[i] for i in sel)).n_records == len(sel)
    ctx.close()
