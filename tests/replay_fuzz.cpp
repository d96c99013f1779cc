// replay_fuzz.cpp — differential fuzz of csrc/replay.h (the host-side replay of the reference's Buffer arithmetic, which
// decides "Fastq record is too long" and cuts RecordSets) against the oracle's streaming restatement, at the
// reference's own fuzzing BUFSIZE of 64 (src/lib.rs:126-127) and at 69632.  TEST INFRASTRUCTURE: built by
// tests/test_replay_fuzz.py with -fsanitize=address,undefined, links oracle/libfqref.so, needs no GPU.
//
// Inputs are valid records of random sizes (many around BUFSIZE), optionally cut short at the end; the record
// boundaries a scan would deliver come from the oracle run with an unlimited buffer; they are fed to BufferReplay in
// random chunkings, as fqh_stream_* does.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../fastq-rs_amd/csrc/replay.h"
extern "C" {
#include "../oracle/fqref.h"
}

static std::string record(std::mt19937_64 &rng, uint64_t total) {  // a valid record of exactly `total` bytes (>= 6)
    // "@" + h + "\n" + s + "\n+\n" + q + "\n" with |s| == |q|: total = 6 + |h| + 2 |s|
    uint64_t body = total - 6, s = (body / 2 > 0) ? rng() % (body / 2 + 1) : 0, h = body - 2 * s;
    std::string r = "@" + std::string(h, 'h') + "\n" + std::string(s, 'A') + "\n+\n" + std::string(s, 'I') + "\n";
    return r;
}

int main(int argc, char **argv) {
    const uint64_t iters = argc > 1 ? strtoull(argv[1], 0, 0) : 2000;
    std::mt19937_64 rng(argc > 2 ? strtoull(argv[2], 0, 0) : 12345);
    uint64_t fails = 0, tripped_cases = 0, short_trips = 0;
    for (uint64_t it = 0; it < iters; ++it) {
        const uint64_t B = (it % 4 == 3) ? 69632 : 64;
        std::string data;
        const int nrec = 1 + (int)(rng() % 40);
        for (int i = 0; i < nrec; ++i) {
            uint64_t len;
            const uint64_t r = rng() % 60;
            if (r < 50) len = 6 + rng() % (B / 2);
            else if (r < 59) len = B - 20 + rng() % 21;    // around the alignment-dependent band (B - 15 .. B)
            else len = B + 1 + rng() % B;                    // certainly too long
            if (len < 6) len = 6;
            data += record(rng, len);
        }
        bool truncated = false;
        uint64_t bad_need = 0;  // != 0: the input ends with an INVALID record whose error shows once bad_need of its bytes are visible
        if (rng() % 5 == 0 && data.size() > 3) {  // cut inside the last record: a truncated tail
            data.resize(data.size() - 1 - rng() % 3);
            truncated = true;
        } else if (rng() % 4 == 0) {  // a broken record behind the valid ones, its error near the end of the window or beyond it
            const uint64_t r = rng() % 10;
            const uint64_t len = r < 7 ? B - 24 + rng() % 30 : 8 + rng() % (2 * B);
            const uint64_t body = len > 8 ? len - 8 : 0, sq = body / 2 > 0 ? rng() % (body / 2 + 1) : 0, h = body - 2 * sq;
            const uint64_t kind = rng() % 3;
            if (kind == 0) {         // separator line without '+': shows with the first byte of that line
                data += "@" + std::string(h, 'h') + "\n" + std::string(sq, 'A') + "\n-\n" + std::string(sq, 'I') + "\n";
                bad_need = 1 + h + 1 + sq + 1 + 1;
            } else if (kind == 1) {  // quality one byte longer than the sequence: shows with the record's last newline
                data += "@" + std::string(h, 'h') + "\n" + std::string(sq, 'A') + "\n+\n" + std::string(sq + 1, 'I') + "\n";
                bad_need = 1 + h + 1 + sq + 1 + 2 + sq + 1 + 1;
            } else {                 // a header that does not start with '@': shows with its first byte
                data += "x" + std::string(h, 'h') + "\n" + std::string(sq, 'A') + "\n+\n" + std::string(sq, 'I') + "\n";
                bad_need = 1;
            }
            if (rng() % 3 == 0) data += record(rng, 6 + rng() % 40);  // (what follows a broken record is never looked at)
        }
        const uint8_t *p = (const uint8_t *)data.data();
        // what the scan delivers: all record boundaries (unlimited buffer), and the status before the too-long rule
        fqref_result big;
        std::vector<uint64_t> rs(data.size() / 6 + 2);
        fqref_offsets(p, data.size(), ((data.size() + 15) & ~(uint64_t)15) + 2 * 69632, 0, rs.data(), rs.size(), &big);  // (a buffer the whole input fits)
        const uint64_t n = big.n_records;
        rs.resize(n + 1);
        rs[n] = big.bytes_consumed;
        // the reference
        fqref_result want;
        fqref_count(p, data.size(), B, 0, &want);
        // the closed form (stateless: fqh::TooLong) over the same boundaries; for the invalid record behind them the bytes it
        // needs, for a truncated tail 0, for a clean end NO_BAD
        {
            uint64_t w3 = 0;
            const uint64_t need3 = bad_need ? bad_need : (truncated ? 0 : fqh::TooLong::NO_BAD);
            const bool t3 = fqh::TooLong::first(B, rs.data(), n, data.size() - rs[n], need3, &w3);
            const bool want3 = want.status == FQREF_E_TOO_LONG;
            bool ok3 = t3 == want3 && (!t3 || w3 == want.n_records);
            if (!t3 && ok3) ok3 = want.n_records == n && want.status == big.status;
            if (!ok3) {
                ++fails;
                if (fails < 5)
                    fprintf(stderr, "CLOSED FORM MISMATCH it=%llu B=%llu n=%llu t3=%d w3=%llu want status=%d n=%llu big status=%d bad_need=%llu\n",
                            (unsigned long long)it, (unsigned long long)B, (unsigned long long)n, t3, (unsigned long long)w3, want.status,
                            (unsigned long long)want.n_records, big.status, (unsigned long long)bad_need);
            }
            // a byte range judged by itself: the records from a random boundary on (one the reference reaches), with their TRUE
            // file offsets, give the same verdict
            if (n > 1) {
                const uint64_t from = rng() % (std::min<uint64_t>(n, want.n_records) + 1);
                uint64_t w4 = 0;
                const bool t4 = fqh::TooLong::first(B, rs.data() + from, n - from, data.size() - rs[n], need3, &w4);
                if (t4 != want3 || (t4 && from + w4 != want.n_records)) {
                    ++fails;
                    if (fails < 5) fprintf(stderr, "CLOSED FORM (RANGE) MISMATCH it=%llu from=%llu\n", (unsigned long long)it, (unsigned long long)from);
                }
            }
        }
        if (bad_need) continue;  // (the replay below is fed valid records and truncated tails only, as before)
        // the replay, fed in random chunks of boundaries
        fqh::BufferReplay rp;
        rp.reset(B);
        uint64_t which = 0, done = 0;
        bool trip = false;
        while (!trip) {
            const uint64_t take = (rng() % 3 == 0) ? n - done : (n - done ? rng() % (n - done + 1) : 0);
            const bool last = done + take == n;
            const uint64_t known_end = last ? data.size() : rs[done + take] + (rng() % 2 ? 0 : (rs[done + take + (done + take < n ? 1 : 0)] - rs[done + take]) / 2);
            const uint64_t need = (last && truncated) ? 0 : fqh::BufferReplay::NO_BAD;
            trip = rp.step(rs.data() + done, done, take, last ? data.size() : known_end, last, need, &which);
            done += take;
            if (last) break;
        }
        const bool want_trip = want.status == FQREF_E_TOO_LONG;
        bool ok = trip == want_trip && (!trip || which == want.n_records);
        if (!trip && ok) ok = want.n_records == n && (want.status == 0 || want.status == FQREF_E_TRUNCATED);
        // RecordSets: sizes of the yielded sets
        std::vector<uint64_t> sizes(data.size() / 8 + 16), got;
        uint64_t nsets = 0;
        fqref_result ws;
        fqref_record_sets(p, data.size(), B, 0, 1, sizes.data(), sizes.size(), &nsets, nullptr, &ws);
        fqh::BufferReplay rq;
        rq.reset(B, true);
        uint64_t w2 = 0;
        bool fin = false;
        const bool t2 = rq.step(rs.data(), 0, n, data.size(), true, truncated ? 0 : fqh::BufferReplay::NO_BAD, &w2, &got, &fin);
        if ((ws.status == FQREF_E_TOO_LONG) != t2) ok = false;
        if (got.size() > nsets) ok = false;  // (a set under construction when an error hits is dropped by the reference)
        for (size_t i = 0; i < got.size() && i < nsets; ++i)
            if (got[i] != sizes[i]) ok = false;
        if (ws.status == 0 && got.size() != nsets) ok = false;
        // ---- a reader that comes back short (the oracle's max_read: at most that many bytes per read() call).  The host's own
        // reads — any sizes asked for, each answered with min(asked, max_read) — are noted; the replay must then walk the
        // reference's sequence of reads: Parser::each's verdict and RecordSetIter's cuts under the same reader.
        {
            // (one case in four: a reader that fills every read — a file, whose last read alone comes back short)
            const uint64_t mr = rng() % 4 == 0 ? data.size() + 1 + rng() % 100 : 1 + rng() % (B == 64 ? 70 : (rng() % 2 ? 4096 : 70000));
            fqref_result wm;
            fqref_count(p, data.size(), B, mr, &wm);
            std::vector<uint64_t> msizes(data.size() + 16), mgot;
            uint64_t mnsets = 0;
            fqref_result wms;
            fqref_record_sets(p, data.size(), B, mr, 1, msizes.data(), msizes.size(), &mnsets, nullptr, &wms);
            for (int mode = 0; mode < 2; ++mode) {
                fqh::BufferReplay rm;
                rm.reset(B, mode == 1);
                for (uint64_t pos = 0; pos < data.size();) {   // the host's reads: slots of at least BUFSIZE bytes, filled read by read
                    const uint64_t slot = B + rng() % (4 * B);
                    for (uint64_t filled = 0; filled < slot && pos < data.size();) {
                        const uint64_t asked = slot - filled;
                        const uint64_t gotb = std::min<uint64_t>(std::min(asked, mr), data.size() - pos);
                        rm.note_read(gotb, asked);
                        pos += gotb;
                        filled += gotb;
                        if (rng() % 7 == 0) break;   // (a host that submits a slot before it is full: Options::low_latency)
                    }
                }
                uint64_t wq = 0, dn = 0;
                bool tq = false, fq = false;
                while (!tq) {   // boundaries in random chunks, as the ring delivers them
                    const uint64_t take = (rng() % 3 == 0) ? n - dn : (n - dn ? rng() % (n - dn + 1) : 0);
                    const bool last = dn + take == n;
                    const uint64_t ke = last ? data.size() : rs[dn + take];
                    tq = rm.step(rs.data() + dn, dn, take, ke, last, (last && truncated) ? 0 : fqh::BufferReplay::NO_BAD, &wq,
                                 mode ? &mgot : nullptr, &fq);
                    dn += take;
                    if (last) break;
                }
                bool okm;
                if (mode == 0) {
                    okm = tq == (wm.status == FQREF_E_TOO_LONG) && (!tq || wq == wm.n_records);
                    if (!tq && okm) okm = wm.n_records == n && (wm.status == 0 || wm.status == FQREF_E_TRUNCATED);
                } else {
                    okm = tq == (wms.status == FQREF_E_TOO_LONG) && mgot.size() <= mnsets && (wms.status != 0 || mgot.size() == mnsets);
                    for (size_t i = 0; i < mgot.size() && i < mnsets; ++i) okm = okm && mgot[i] == msizes[i];
                }
                if (!okm) {
                    ++fails;
                    if (fails < 5)
                        fprintf(stderr, "SHORT-READ MISMATCH it=%llu B=%llu max_read=%llu mode=%d trip=%d which=%llu want status=%d n=%llu sets %zu / %llu\n",
                                (unsigned long long)it, (unsigned long long)B, (unsigned long long)mr, mode, tq, (unsigned long long)wq,
                                mode ? wms.status : wm.status, (unsigned long long)wm.n_records, mgot.size(), (unsigned long long)mnsets);
                }
                short_trips += tq;
            }
        }
        tripped_cases += trip;
        if (!ok) {
            ++fails;
            if (fails < 5)
                fprintf(stderr, "MISMATCH it=%llu B=%llu n=%llu trip=%d which=%llu want status=%d n=%llu sets got=%zu want=%llu t2=%d wsstatus=%d\n",
                        (unsigned long long)it, (unsigned long long)B, (unsigned long long)n, trip, (unsigned long long)which, want.status,
                        (unsigned long long)want.n_records, got.size(), (unsigned long long)nsets, t2, ws.status);
        }
    }
    printf("replay_fuzz: %llu inputs, %llu too-long cases (%llu more under short reads), %llu mismatches\n", (unsigned long long)iters,
           (unsigned long long)tripped_cases, (unsigned long long)short_trips, (unsigned long long)fails);
    return fails ? 1 : 0;
}
