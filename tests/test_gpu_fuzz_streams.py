"""Differential fuzz of the CHUNKED and STREAMED statistics: random files (clean and dirty, ragged, CRLF, damaged, truncated, reads
of 36 bp to 2 kbp) through FQH_STREAM_STATS rings of random slot sizes — with short submits — and through fqh_stats_launch_lead over
random chunks, against the oracle: status, record count, scalars, histograms.  Shakes the single pass with a carry (k_scan_stats +
k_stats_edge), its deferred commit in the ring, the scans enqueued ahead of their collect, and every fall-back to the two-pass
route (the reference touches a record once whatever the buffer: src/lib.rs:226-237, src/buffer.rs:51-100).  Round 3 found three
bugs with it (a chunk whose last group holds no line start, a rerun of the fast path that committed early, a slot refilled under
a pending commit).  tools/fuzz_streams.py runs the same function for as long as one likes."""
import ctypes as C
import time

import numpy as np
import pytest

import fuzzgen

pytestmark = pytest.mark.gpu


def fuzz_streams(torch, pkg, fqref, seed, budget, max_cases=None):
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    ALPH = np.frombuffer(b"ACGTN", dtype=np.uint8)
    def clean(nrec, L, crlf, ragged):
        out = []
        for i in range(nrec):
            n = int(rng.integers(max(0, L - ragged), L + 1)) if ragged else L
            e = b"\r\n" if (crlf and rng.random() < 0.5) else b"\n"
            seq = rng.choice(ALPH, n, p=[.2475, .2475, .2475, .2475, .01]).tobytes()
            qual = rng.integers(33, 75, n).astype(np.uint8).tobytes()
            out.append(b"@r%d x" % i + e + seq + e + b"+" + e + qual + e)
        return b"".join(out)
    t_end = time.time() + budget
    cases = single = 0
    while time.time() < t_end and (max_cases is None or cases < max_cases):
        L = int(rng.choice([36, 75, 100, 150, 151, 180, 250, 300, 320, 400, 500, 511, 520, 2000]))
        kind = rng.random()
        if kind < 0.6:
            data = clean(int(rng.integers(500, 1 + (6 << 20) // (2 * L + 20))), L, rng.random() < 0.2, int(rng.choice([0, 0, 5, 60])))
        else:
            data = fuzzgen.valid_file(rng, int(rng.integers(200, 6000)), maxlen=L)
        if rng.random() < 0.2:
            data = fuzzgen.mutate(rng, data, 1)
        if rng.random() < 0.1:
            data = data[: len(data) - int(rng.integers(1, 200))]
        lmax = int(rng.choice([36, 64, 100, 128, 150, 192, 256, 300, 320, 384, 500, 512, 600]))
        r, oq, ob, osc = fqref.stats(data, lmax)
        a = np.frombuffer(data, dtype=np.uint8)
        qh = torch.zeros(lmax * 256, dtype=torch.int64, device=dev); bh = torch.zeros(lmax * 8, dtype=torch.int64, device=dev)
        sc = torch.zeros(8, dtype=torch.int64, device=dev)
        ctx = pkg.Ctx(0)
        if rng.random() < 0.5:   # ---- the ring
            slot = int(rng.choice([4096, 65536, 300000 // 16 * 16, 1 << 20, 3 << 20]))
            # round 6: slots fed from the caller's own memory (fqh_stream_submit_external; "external": a ring without pinned data
            # slots, "mixed": either way per slot), chunks HELD and given back out of order (fqh_stream_release_chunk), and rings
            # made of a parked ring's memory (FQH_OPT_KEEP_RING: the context's second ring of the case)
            feed = str(rng.choice(["pinned", "pinned", "external", "mixed"]))
            n_slots = int(rng.integers(2, 5))
            hold = int(rng.integers(0, n_slots - 1)) if rng.random() < 0.4 else 0   # chunks kept back besides the current one
            keep = rng.random() < 0.3
            if keep:
                ctx.set_keep_ring(True)
                pkg.Stream(ctx, slot, n_slots, pkg.STREAM_STATS | (pkg.STREAM_EXTERNAL if feed == "external" else 0)).close()
            st = pkg.Stream(ctx, slot, n_slots, pkg.STREAM_STATS | (pkg.STREAM_EXTERNAL if feed == "external" else 0))
            st.set_stats(lmax, qh.data_ptr(), bh.data_ptr(), sc.data_ptr())
            src = a.copy() if len(a) else np.zeros(1, np.uint8)
            if feed != "pinned" and rng.random() < 0.7:
                ctx.host_register(src.ctypes.data, max(1, len(data)))
                registered = True
            else:
                registered = False
            pos = nrec = sub = col = 0
            status, done = pkg.OK, False
            held = []
            while True:
                while not done and sub - col + len(held) < n_slots:
                    ext = feed == "external" or (feed == "mixed" and rng.random() < 0.5)
                    room = slot
                    if not ext:
                        acq = st.acquire()
                        if acq is None: break
                        room = acq[1]
                    n = min(room, len(data) - pos)
                    if rng.random() < 0.3: n = min(n, int(rng.integers(1, room + 1)))
                    last = pos + n >= len(data)
                    if ext:
                        if not st.submit_external(src.ctypes.data + pos, n, last): break
                    else:
                        C.memmove(acq[0], data[pos: pos + n], n)
                        st.submit(n, last)
                    pos += n; done = last
                    sub += 1
                if col == sub: break
                c = st.collect()
                if c is None:            # (FQH_E_AGAIN: the slot behind the one to collect is still held)
                    assert held, (seed, cases, "E_AGAIN with nothing held")
                    st.release_chunk(held.pop(int(rng.integers(0, len(held)))))
                    continue
                col += 1
                nrec += c.n_records
                single += bool(ctx.last_scan_fast())
                held.append(c)
                while len(held) > hold:  # give back in a random order
                    st.release_chunk(held.pop(int(rng.integers(0, len(held)))))
                if c.parse_status != pkg.OK: status = c.parse_status; break
                if c.is_final: break
            for c in held:
                st.release_chunk(c)
            torch.cuda.synchronize()
            st.close()
            if registered:
                ctx.host_unregister(src.ctypes.data)
            what = ("ring", slot, feed, n_slots, hold, keep)
        else:                    # ---- hand-made chunks of one device buffer, with the lead in front of each
            n = len(data)
            d = torch.empty(n + 64, dtype=torch.uint8, device=dev); d[:n].copy_(torch.from_numpy(a.copy()))
            cuts = [0] + sorted(int(x) // 16 * 16 for x in rng.integers(1, max(2, n), int(rng.integers(1, 6)))) + [n]
            carry, nrec, status = None, 0, pkg.OK
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                if lo == hi: continue
                ctx.stats_launch_lead(d.data_ptr() + lo, hi - lo, lo, lmax, qh.data_ptr(), bh.data_ptr(), sc.data_ptr(), is_final=(hi == n), carry=carry)
                s, carry = ctx.stats_finish()
                nrec += s.n_records
                single += bool(ctx.last_scan_fast())
                if s.parse_status != pkg.OK: status = s.parse_status; break
            what = ("chunks", cuts)
        ctx.close()
        assert (status, nrec) == (r.status, r.n_records), (seed, cases, what, (status, nrec), (r.status, r.n_records))
        assert np.array_equal(sc.cpu().numpy().astype(np.uint64), osc), (seed, cases, what, sc.cpu().numpy(), osc)
        assert np.array_equal(qh.cpu().numpy().astype(np.uint64).reshape(lmax, 256), oq), (seed, cases, what, "qual")
        assert np.array_equal(bh.cpu().numpy().astype(np.uint64).reshape(lmax, 8), ob), (seed, cases, what, "base")
        cases += 1
    return cases, single


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fuzz_streams(fqref, seed):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as g
    pkg = g.load_package()
    cases, single = fuzz_streams(torch, pkg, fqref, seed, 20.0, max_cases=120)
    assert cases >= 20 and single >= 20, (cases, single)
