"""tools/make_shard_fixture.py — run ON A GPU: the words the sharded modes exchange, as the HIP kernels produce them, dumped to
tests/golden/shard_words_gpu.json for the CPU tests (tests/test_shard_fixture.py, the gloo workers of tests/test_shard_carry.py
and tests/test_shard_keys.py): the folds and reductions of the N > 1 path are then driven by REAL kernel output, not by
numpy-derived stand-ins.  The input files are regenerated on the CPU side from (seed, parameters) with tests/fuzzgen.py; the
fixture keeps their SHA-256.  A fixture is data: words in, expected outcomes (the oracle's) next to them."""
import ctypes as C, hashlib, importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import __graft_entry__ as g, fuzzgen
from oracle import fqref
pkg = g.load_package()
sharded = importlib.import_module("fastq_rs_amd.sharded")
dev = torch.device("cuda:0")


def make_file(spec):
    rng = np.random.default_rng(spec["seed"])
    data = bytearray(fuzzgen.valid_file(rng, spec["records"], maxlen=spec["maxlen"], crlf=False))
    for off, byte in spec.get("patch", []):
        data[off] = byte
    return bytes(data)


def hbm_case(spec, cuts):
    data = make_file(spec)
    n = len(data)
    bounds = [0] + cuts + [n]
    a = np.frombuffer(data, dtype=np.uint8)
    rows, rescans = [], []
    carry = None
    for r in range(len(bounds) - 1):
        lo, hi = bounds[r], bounds[r + 1]
        d = torch.zeros(max(16, hi - lo + 16), dtype=torch.uint8, device=dev)
        d[: hi - lo].copy_(torch.from_numpy(a[lo:hi].copy()))
        ctx = pkg.Ctx(0)
        nn, ns, back0 = ctx.shard_prescan(d.data_ptr(), hi - lo)
        rows.append([hi - lo, nn, ns] + back0)
        rs = torch.zeros((hi - lo) // 6 + 16, dtype=torch.int64, device=dev)
        ctx.rescan_launch(r == len(bounds) - 2, carry, rs.data_ptr(), rs.numel())
        s, c, st = ctx.scan_finish()
        rescans.append({"n_records": int(s.n_records), "status": int(s.parse_status), "err_record": int(s.err_record)})
        carry = pkg.carry_combine(carry, *rows[-1][:3], rows[-1][3:])
        ctx.close()
    r = fqref.count(data)
    return {"file": spec, "sha256": hashlib.sha256(data).hexdigest(), "len": n, "cuts": cuts, "prescan_words": rows, "rescan": rescans,
            "oracle": {"status": int(r.status), "n_records": int(r.n_records)}}


def stream_case(spec, cuts, lmax=64):
    data = make_file(spec)
    n = len(data)
    host = (C.c_uint8 * n).from_buffer_copy(data)

    def read_into(addr, off, nbytes):
        C.memmove(addr, C.addressof(host) + off, nbytes)

    bounds = [0] + cuts + [n]
    world = len(bounds) - 1
    shards, hists = [], []
    for r in range(world):
        h = torch.zeros(8 + lmax * 264, dtype=torch.int64, device=dev)
        hists.append(h)
        ctx = pkg.Ctx(0)
        shards.append(sharded.stream_shard(ctx, read_into, bounds[r], bounds[r + 1], n, 1 << 16,
                                           stats=(lmax, h[8: 8 + lmax * 256].data_ptr(), h[8 + lmax * 256:].data_ptr(), h[:8].data_ptr())))
        ctx.close()
    words = [sh.words() for sh in shards]
    fin = []
    for r in range(world):
        h = hists[r]
        ctx = pkg.Ctx(0)
        fin.append(list(sharded.finish(ctx, read_into, n, words, r, 1 << 16,
                                       stats=(lmax, h[8: 8 + lmax * 256].data_ptr(), h[8 + lmax * 256:].data_ptr(), h[:8].data_ptr()))))
        ctx.close()
    status, n_records, err_offset = sharded.outcome([f[0] for f in fin], min(f[1] for f in fin))
    r, oq, ob, osc = fqref.stats(data, lmax)
    assert (status, n_records) == (r.status, r.n_records), (spec, cuts, status, n_records, r.status, r.n_records)
    tot = sum(h.cpu().numpy().astype(np.uint64) for h in hists)
    if r.status == 0:
        assert np.array_equal(tot[:8], osc)
    return {"file": spec, "sha256": hashlib.sha256(data).hexdigest(), "len": n, "cuts": cuts, "lmax": lmax, "words": words, "finish": fin,
            "outcome": [status, n_records, err_offset], "oracle": {"status": int(r.status), "n_records": int(r.n_records)},
            "scalars_sum": [int(x) for x in tot[:8]]}


def boundary_cut(data, frac):
    at = data.index(b"\n@r", int(len(data) * frac)) + 1   # a record start (headers are "@r<i>...")
    return at


out = {"made_by": "tools/make_shard_fixture.py on an MI355X (fqh_shard_prescan / fqh_rescan_launch / fqh_shard_stream_run / fqh_shard_stream_finish)"}
spec = {"seed": 424242, "records": 6000, "maxlen": 120}
data = make_file(spec)
n = len(data)
cuts8 = [n * k // 8 + 13 * k for k in range(1, 8)]
out["hbm_ranks8"] = hbm_case(spec, cuts8)
out["hbm_ranks2"] = hbm_case(spec, [n // 2 + 7])
out["stream_ranks8"] = stream_case(spec, cuts8)
out["stream_ranks2_cut_in_record"] = stream_case(spec, [n // 2 + 7])
out["stream_ranks2_cut_at_record_start"] = stream_case(spec, [boundary_cut(data, 0.5)])
# an error in rank 5 of 8: the separator line of the first record that starts behind 5/8 + 1000 loses its '+'
k = data.index(b"\n+\n", n * 5 // 8 + 1000) + 1
bad = dict(spec, patch=[[k, ord("-")]])
out["stream_ranks8_error_in_rank5"] = stream_case(bad, cuts8)
out["hbm_ranks8_error_in_rank5"] = hbm_case(bad, cuts8)
# ... and one in the record that straddles the cut between ranks 4 and 5
k2 = data.rindex(b"\n+", 0, cuts8[4]) + 1
bad2 = dict(spec, patch=[[k2, ord("x")]])
out["stream_ranks8_error_across_cut"] = stream_case(bad2, cuts8)
path = os.path.join(ROOT, "tests", "golden", "shard_words_gpu.json")
with open(path, "w") as f:
    json.dump(out, f, indent=1)
print("wrote", path, {k: v.get("outcome", v.get("oracle")) for k, v in out.items() if isinstance(v, dict)})
