// shard_stream.hip — the byte-range sharded, host-streamed mode (BASELINE.json configs[4]) under the C ABI:
// fqh_shard_stream_run / fqh_shard_stream_finish / fqh_shard_stream_outcome.
//
// The reference's analogue is Parser::parallel_each (src/lib.rs:509-565): per-worker results are gathered at the end
// (src/lib.rs:553-559) and a parse error is what the whole call returns (src/lib.rs:544-547, 561-564) — THE error of the
// sequential parse, kind and record.  Here the file is cut at arbitrary byte offsets, one rank = one GPU = one pinned ring; a
// rank cannot wait for the ranks in front of it (they stream for seconds), so it works PHASE-FREE and the ranks talk once:
//   1. rank r > 0 uploads a few MiB from the start of its range and asks fqh_shard_align for the line phase (newlines in
//      front of the range, mod 4) and for the offset R of its first record: the one phase under which the window parses.
//      It becomes an ANCHOR: it streams [lo + R, hi) through fqh_stream_* like a file of its own that begins at file offset
//      lo + R (every record validated in the reference's order, src/records.rs:201-247; "too long" judged on true file offsets,
//      csrc/replay.h; histograms added on the way).  A range in which no record starts (PASS), or whose window does not single
//      out a phase (DEFER), streams nothing and only counts its newlines;                              -> fqh_shard_stream_run
//   2. one all-gather of FQH_SHARD_STREAM_WORDS words per rank (no bytes: whoever needs bytes of another rank's range reads
//      them through its own `read` callback);
//   3. every rank derives the same picture from the words: the TRUE newline count in front of every range, hence which
//      anchors parsed under the true phase (TRUSTED: validity under the true line phase is what the sequential parser computes,
//      DESIGN.md section 2).  Between the last complete record of one trusted anchor and the first record of the next lies a
//      GAP — the record that straddles the cut, plus every PASS / DEFER / untrusted range in between.  The anchor a gap ends at
//      parses it: a sequential parse from a TRUE record start over the file's own bytes, so kind and record of an error in it
//      are the reference's; it must land exactly on the anchor's first record.  What lies behind the last trusted anchor is
//      parsed to the end of the file by the last rank that holds bytes.  Every rank packs its first error as
//      (file offset of the failing record, rank, kind) into one u64 key;                               -> fqh_shard_stream_finish
//   4. one all-reduce SUM of [records per rank, scalars, histograms] and one all-reduce MIN of the keys: the minimum is the
//      first error in FILE order — the error Parser::each returns — and the records delivered before it are the sum of the
//      slots up to its rank (fqh_shard_stream_outcome); or the totals.
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "ctx.h"

namespace {
constexpr uint64_t ALIGN_WINDOW = 4ull << 20;
// A range up to this size always reports the exact newline count of ALL its bytes, even when its stream stopped at an error:
// its alignment window reaches behind its end, so an error the window saw need not lie inside the range (see classify()).
constexpr uint64_t SMALL_RANGE = ALIGN_WINDOW + 4ull * FQH_BUFSIZE;
enum : int { W_STATUS, W_RECORDS, W_NEWLINES, W_PHASE, W_HEAD, W_TAIL, W_ERR_OFFSET, W_FLAGS, W_LO, W_HI };
constexpr uint64_t FLAG_NL_INCOMPLETE = 1;  // W_NEWLINES does not cover the whole range (the stream stopped at an error in a large range)
constexpr uint64_t FLAG_IO_FAILED = 2;      // the read callback failed for bytes of this range: the range parsed nothing by itself (DEFER,
                                            // with FLAG_NL_INCOMPLETE), what it counted is void; whoever parses the open gap it
                                            // lies in reads its bytes again IN FILE ORDER and meets a parse error in front of
                                            // the unreadable bytes, or the failure

uint64_t pack_key(uint64_t offset, int rank, int32_t status) {
    uint64_t code;
    if (status >= FQH_E_HEADER && status <= FQH_E_TOO_LONG) code = (uint64_t)(status - 1);
    else if (status == FQH_E_IO) code = 5;
    else if (status == FQH_E_CAPACITY) code = 6;
    else code = 7;
    return (offset << 11) | ((uint64_t)(rank & 0xFF) << 3) | code;
}

fqh_status count_newlines(fqh_ctx *ctx, fqh_read_fn read, void *user, uint64_t from, uint64_t to, uint64_t *n) {
    std::vector<uint8_t> buf(std::min<uint64_t>(to - from, 4ull << 20));
    for (uint64_t at = from; at < to;) {
        const uint64_t k = std::min<uint64_t>(buf.size(), to - at);
        if (read(user, buf.data(), at, k) != 0) {
            if (ctx) ctx->err = "fqh_shard_stream: the read callback failed";
            return FQH_E_IO;
        }
        for (const uint8_t *p = buf.data(), *e = p + k; (p = (const uint8_t *)memchr(p, '\n', (size_t)(e - p))) != nullptr; ++p) ++*n;
        at += k;
    }
    return FQH_OK;
}

struct Span {
    int32_t status = FQH_OK;       // first parse error (FQH_OK: none)
    uint64_t n_records = 0;        // records delivered before it
    uint64_t err_offset = 0;       // file offset of the failing record
    uint64_t end_of_records = 0;   // file offset behind the last delivered record
    uint64_t n_newlines = 0;       // '\n' in the bytes the stream has looked at ...
    uint64_t seen_to = 0;          // ... which are [from, seen_to)
};

// Parses the file's bytes [from, to) — `from` is a record start — through a pinned ring, exactly as a file of its own that
// begins at file offset `from` (fqh_stream_set_origin: offsets in the result are file offsets, and so is what "too long" is
// judged on).  is_final: `to` is the end of the file (EOF rule, src/lib.rs:264-294).
// map != NULL: the bytes are taken in place from the caller's page-locked memory (fqh_stream_submit_external), no pinned slots.
fqh_status stream_span(fqh_ctx *ctx, fqh_read_fn read, fqh_map_fn map, void *user, uint64_t from, uint64_t to, bool is_final,
                       uint64_t slot_bytes, uint32_t n_slots, uint32_t lmax, uint64_t *d_qual_hist, uint64_t *d_base_hist,
                       uint64_t *d_scalars, Span *out) {
    *out = Span{};
    out->end_of_records = out->seen_to = from;
    if (to <= from && !is_final) return FQH_OK;
    // (a gap of a few hundred bytes does not need the caller's ring of many MiB: pinned memory is slow to come by)
    const uint64_t want = std::max<uint64_t>(1u << 16, ((to - from) + 15) & ~(uint64_t)15);
    const uint64_t sb = std::min<uint64_t>(slot_bytes, want);
    fqh_stream *sp = nullptr;
    const bool stats = lmax != 0;
    fqh_status st = fqh_stream_create(ctx, sb, n_slots, (stats ? FQH_STREAM_STATS : 0u) | (map ? FQH_STREAM_EXTERNAL : 0u), &sp);
    if (st != FQH_OK) return st;
    struct Closer {
        fqh_stream *s;
        ~Closer() { fqh_stream_destroy(s); }
    } closer{sp};
    st = fqh_stream_set_origin(sp, from);
    if (st != FQH_OK) return st;
    if (stats) {
        st = fqh_stream_set_stats(sp, lmax, d_qual_hist, d_base_hist, d_scalars);
        if (st != FQH_OK) return st;
    }
    uint64_t pos = from;
    bool done_reading = false, io_failed = false;
    uint64_t submitted = 0, collected = 0;
    for (;;) {
        while (!done_reading && !io_failed && map) {
            if (submitted - collected >= n_slots) break;  // the ring is full: collect first
            uint64_t n = std::min<uint64_t>(sb, to - pos), avail = 0;
            const uint8_t *src = n ? map(user, pos, n, &avail) : nullptr;
            if (n && (!src || !avail)) {
                io_failed = true;  // (file order: see below)
                break;
            }
            n = std::min<uint64_t>(n, avail);
            done_reading = pos + n >= to;
            st = fqh_stream_submit_external(sp, src, n, (done_reading && is_final) ? 1 : 0);
            if (st != FQH_OK) return st;
            pos += n;
            ++submitted;
        }
        while (!done_reading && !io_failed && !map) {
            uint8_t *dst = nullptr;
            uint64_t cap = 0;
            st = fqh_stream_acquire(sp, &dst, &cap);
            if (st == FQH_E_CAPACITY) break;  // the ring is full: collect first
            if (st != FQH_OK) return st;
            const uint64_t n = std::min<uint64_t>(cap, to - pos);
            if (n && read(user, dst, pos, n) != 0) {
                // file order: a parse error in the slots already on their way lies in FRONT of the bytes that could not be
                // read, and the sequential reader would have met it first — collect those before giving up
                io_failed = true;
                break;
            }
            pos += n;
            done_reading = pos >= to;
            st = fqh_stream_submit(sp, n, (done_reading && is_final) ? 1 : 0);
            if (st != FQH_OK) return st;
            ++submitted;
        }
        if (collected == submitted) break;
        fqh_chunk c;
        st = fqh_stream_collect(sp, &c);
        if (st != FQH_OK) return st;
        ++collected;
        out->n_records += c.n_records;
        out->end_of_records = c.h_rec_start[c.n_records];
        out->seen_to = c.base_offset + c.data_len;
        if (c.parse_status != FQH_OK) {
            out->status = c.parse_status;
            out->err_offset = c.err_offset;
            (void)fqh_stream_release(sp);
            break;
        }
        st = fqh_stream_release(sp);
        if (st != FQH_OK) return st;
    }
    if (io_failed && out->status == FQH_OK) {
        ctx->err = "fqh_shard_stream: the read callback failed";
        return FQH_E_IO;
    }
    fqh_carry cy;
    st = fqh_stream_carry(sp, &cy);
    if (st != FQH_OK) return st;
    out->n_newlines = cy.nl_count;
    return FQH_OK;
}

// What every rank derives from the gathered words, the same on all of them: who parsed under the true line phase, and which
// rank parses which gap.
struct Job {
    bool gap = false;        // this rank parses the file's bytes [gap_from, gap_to) ...
    bool gap_final = false;  // ... to the end of the file (EOF rule), or up to its own first record, where the parse must land
    uint64_t gap_from = 0, gap_to = 0;
    bool own = false;        // its own streamed records count (a trusted anchor)
    bool failed = false;     // it could not do its part (an I/O or device error): key at fail_offset
    uint64_t fail_offset = 0;
};

bool classify(const uint64_t *all, int n_ranks, int rank, uint64_t file_len, Job *job) {
    auto W = [&](int j) { return all + (size_t)j * FQH_SHARD_STREAM_WORDS; };
    *job = Job{};
    int last_nonempty = -1;
    for (int j = 0; j < n_ranks; ++j)
        if (W(j)[W_PHASE] != FQH_SHARD_EMPTY || (int32_t)W(j)[W_STATUS] > FQH_E_TOO_LONG) last_nonempty = j;
    uint64_t nl_before = 0;
    // A trusted anchor without an error always lies in front: the start of the file is one (phase 0, a record starts there).
    uint64_t S = 0;            // the open gap begins behind that anchor's last complete record, at this file offset
    bool dead = false;         // a trusted anchor (or a gap in front of one) holds an error of its own: nothing behind it matters
    bool poisoned = false;     // the true newline count is unknown from here on — an error lies in front, inside the open gap:
                               // nobody behind is an anchor any more, the open gap runs to the end of the file
    for (int j = 0; j < n_ranks && !dead; ++j) {
        const uint64_t *w = W(j);
        const int32_t status = (int32_t)w[W_STATUS];
        const uint32_t phase = (uint32_t)w[W_PHASE];
        if (status > FQH_E_TOO_LONG || status < 0) {  // this rank failed locally: the sequential reader would have failed here at the latest
            if (j == rank) {
                job->failed = true;
                job->fail_offset = S;
            }
            dead = true;
            break;
        }
        if (phase == FQH_SHARD_EMPTY) continue;
        const bool anchor = phase <= 3;
        if (anchor && !poisoned) {
            const uint64_t A = w[W_LO] + w[W_HEAD];
            const bool trusted = w[W_LO] == 0 || phase == (uint32_t)(nl_before & 3);
            if (trusted) {
                if (j == rank) {
                    job->own = true;
                    if (S < A) {
                        job->gap = true;
                        job->gap_from = S;
                        job->gap_to = A;
                    }
                }
                if (status != FQH_OK) {
                    dead = true;
                    break;
                }
                S = w[W_HI] - w[W_TAIL];
            }
        }
        // (an anchor that parsed under a wrong phase, a PASS or a DEFER range: bytes of the open gap)
        if (w[W_FLAGS] & FLAG_NL_INCOMPLETE) poisoned = true;
        nl_before += w[W_NEWLINES];
    }
    if (!dead && S < file_len && rank == last_nonempty && !job->own) {
        job->gap = true;
        job->gap_final = true;
        job->gap_from = S;
        job->gap_to = file_len;
    }
    return true;
}
}  // namespace

extern "C" {

void fqh_shard_result_words(const fqh_shard_result *r, uint64_t lo, uint64_t hi, uint64_t words[FQH_SHARD_STREAM_WORDS]) {
    if (!r || !words) return;
    words[W_STATUS] = (uint64_t)(uint32_t)r->status;
    words[W_RECORDS] = r->n_records;
    words[W_NEWLINES] = r->n_newlines;
    words[W_PHASE] = r->phase;
    words[W_HEAD] = r->head_len;
    words[W_TAIL] = r->tail_len;
    words[W_ERR_OFFSET] = r->err_offset;
    words[W_FLAGS] = r->flags;
    words[W_LO] = lo;
    words[W_HI] = hi;
}

void fqh_shard_failed_words(fqh_status why, uint64_t lo, uint64_t hi, uint64_t words[FQH_SHARD_STREAM_WORDS]) {
    if (!words) return;
    for (int i = 0; i < FQH_SHARD_STREAM_WORDS; ++i) words[i] = 0;
    words[W_STATUS] = (uint64_t)(uint32_t)((int32_t)why > FQH_E_TOO_LONG ? why : FQH_E_DEVICE);
    words[W_PHASE] = FQH_SHARD_EMPTY;
    words[W_LO] = lo;
    words[W_HI] = hi;
}

uint64_t fqh_shard_failure_key(int rank, uint64_t offset, fqh_status why) {
    return pack_key(offset, rank, (int32_t)why > FQH_E_TOO_LONG ? (int32_t)why : FQH_E_DEVICE);
}

fqh_status fqh_shard_stream_outcome(uint64_t key, const uint64_t *records_per_rank, int n_ranks, int32_t *status, uint64_t *n_records,
                                    uint64_t *err_offset) {
    if (!status || !n_records || !records_per_rank || n_ranks < 1 || n_ranks > FQH_SHARD_MAX_RANKS) return FQH_E_ARG;
    int upto = n_ranks - 1;
    *status = FQH_OK;
    if (err_offset) *err_offset = 0;
    if (key != FQH_NO_ERROR_KEY) {
        static const int32_t of_code[8] = {FQH_E_HEADER, FQH_E_SEP, FQH_E_LEN_MISMATCH, FQH_E_TRUNCATED, FQH_E_TOO_LONG, FQH_E_IO,
                                           FQH_E_CAPACITY, FQH_E_DEVICE};
        *status = of_code[key & 7u];
        upto = (int)((key >> 3) & 0xFFu);
        if (upto >= n_ranks) return FQH_E_ARG;
        if (err_offset) *err_offset = key >> 11;
    }
    uint64_t n = 0;
    for (int j = 0; j <= upto; ++j) n += records_per_rank[j];
    *n_records = n;
    return FQH_OK;
}

fqh_status fqh_shard_stream_run(fqh_ctx *ctx, fqh_read_fn read, void *user, uint64_t lo, uint64_t hi, uint64_t file_len,
                                uint64_t slot_bytes, uint32_t n_slots, uint32_t lmax, uint64_t *d_qual_hist,
                                uint64_t *d_base_hist, uint64_t *d_scalars, fqh_shard_result *res) {
    return fqh_shard_stream_run_mapped(ctx, read, nullptr, user, lo, hi, file_len, slot_bytes, n_slots, lmax, d_qual_hist, d_base_hist,
                                       d_scalars, res);
}

fqh_status fqh_shard_stream_run_mapped(fqh_ctx *ctx, fqh_read_fn read, fqh_map_fn map, void *user, uint64_t lo, uint64_t hi,
                                       uint64_t file_len, uint64_t slot_bytes, uint32_t n_slots, uint32_t lmax,
                                       uint64_t *d_qual_hist, uint64_t *d_base_hist, uint64_t *d_scalars, fqh_shard_result *res) {
    if (!ctx || !read || !res || hi < lo || hi > file_len || n_slots < 2) return FQH_E_ARG;
    const bool stats = lmax != 0;
    if (stats && (!d_qual_hist || !d_base_hist || !d_scalars)) return FQH_E_ARG;
    *res = fqh_shard_result{};
    res->status = FQH_OK;
    if (hi == lo) {  // an empty range: the rank takes part in the exchange and contributes nothing
        res->phase = FQH_SHARD_EMPTY;
        return FQH_OK;
    }
    uint64_t R = 0;
    if (lo > 0) {
        // ---- where does this range's first record begin, and at which line phase does the range start?  The window only
        // settles that, so it may reach BEHIND the range: a few lines settle nothing — worse, the phase that "gets furthest"
        // in them can be the wrong one (a quality line that starts with '@' right behind the cut looks like a header) — while
        // up to 4 MiB of the file behind lo do, whatever the range's own size.
        uint64_t w = std::min<uint64_t>(ALIGN_WINDOW, file_len - lo);
        std::vector<uint8_t> hostw(w + 1);
        // (one byte more in front: is it a newline?  In pieces: the window reaches behind the range, and bytes there that
        // cannot be read are not this range's failure — the window ends in front of them)
        uint64_t got = 0;
        while (got < w + 1) {
            const uint64_t k = std::min<uint64_t>(1u << 16, w + 1 - got);
            if (read(user, hostw.data() + got, lo - 1 + got, k) != 0) break;
            got += k;
        }
        if (got < std::min<uint64_t>(w, hi - lo) + 1) {
            // bytes of the range itself cannot be read: see FLAG_IO_FAILED
            res->phase = FQH_SHARD_DEFER;
            res->flags |= FLAG_NL_INCOMPLETE | FLAG_IO_FAILED;
            return FQH_OK;
        }
        w = got - 1;
        const uint64_t own = std::min<uint64_t>(w, hi - lo);
        const bool prev_nl = hostw[0] == '\n';
        auto newlines_in = [&](uint64_t n) {
            uint64_t c = 0;
            for (uint64_t i = 0; i < n; ++i) c += hostw[1 + i] == '\n';
            return c;
        };
        // a range that streams nothing still says how many newlines it holds (all of them: the ranks behind it derive their
        // true line phase from the sum) — from the window's bytes, and through the callback for what lies behind them
        auto hand_on = [&](uint32_t kind, bool count_all) -> fqh_status {
            res->phase = kind;
            res->n_newlines = newlines_in(own);
            if (own < hi - lo) {
                const fqh_status cs = count_all ? count_newlines(ctx, read, user, lo + own, hi, &res->n_newlines) : FQH_E_IO;
                if (cs == FQH_E_IO) res->flags |= FLAG_NL_INCOMPLETE | (count_all ? FLAG_IO_FAILED : 0);
                else if (cs != FQH_OK) return cs;
            }
            return FQH_OK;
        };
        bool any_line_start = false;  // (a line starts at lo + i iff the byte in front of it is a newline)
        for (uint64_t i = 0; i < own && !any_line_start; ++i) any_line_start = hostw[i] == '\n';
        if (!any_line_start && own == hi - lo)
            return hand_on(FQH_SHARD_PASS, true);  // not one line starts inside the range, so no record does: it lies inside one line
        uint32_t phase = 0;
        fqh_status st;
        {
            void *win = nullptr;
            st = fqh_dev_alloc(ctx, w + 16, &win);
            if (st != FQH_OK) return st;
            st = fqh_memcpy_h2d(ctx, win, hostw.data() + 1, w);
            if (st == FQH_OK) st = fqh_shard_align(ctx, (const uint8_t *)win, w, prev_nl ? 1 : 0, &phase, &R);
            (void)fqh_dev_free(ctx, win);
        }
        // FQH_E_ARG: several phases validate the window (a few lines at the end of the file; 4 MiB of lines that all start
        // with '@' or '+'); FQH_E_HEADER: none does and none stands out (a parse error at the window's start, a line of
        // megabytes, not FASTQ).  Either way this range does not parse anything by itself (DEFER): after the exchange the
        // true phase is known, and the rank that parses the gap this range lies in reads its bytes under it.  The ranks behind
        // need its newline count: exact when several phases validate (the file may well be valid), and for a range the window
        // covers; a LARGE range whose window validates under no phase holds a parse error under the true one too, inside
        // the window, and whatever lies behind that error is never looked at.
        if (st == FQH_E_ARG) return hand_on(FQH_SHARD_DEFER, true);
        if (st == FQH_E_HEADER) return hand_on(FQH_SHARD_DEFER, hi - lo <= SMALL_RANGE);
        if (st != FQH_OK) return st;
        if (R >= hi - lo) return hand_on(FQH_SHARD_PASS, true);  // no record starts inside the range: the record in progress runs THROUGH it
        res->phase = phase;
        res->head_len = R;
        res->n_newlines = newlines_in(R);
    }
    // ---- an anchor: [lo + R, hi) as a file of its own
    Span sp;
    fqh_status st = stream_span(ctx, read, map, user, lo + R, hi, hi >= file_len, slot_bytes, n_slots, lmax, d_qual_hist, d_base_hist,
                                d_scalars, &sp);
    if (st == FQH_E_IO) {  // (no parse error in what could be read: see FLAG_IO_FAILED)
        *res = fqh_shard_result{};
        res->status = FQH_OK;
        res->phase = FQH_SHARD_DEFER;
        res->flags = FLAG_NL_INCOMPLETE | FLAG_IO_FAILED;
        return FQH_OK;
    }
    if (st != FQH_OK) return st;
    res->status = sp.status;
    res->n_records = sp.n_records;
    res->n_newlines += sp.n_newlines;
    if (sp.status != FQH_OK) {
        res->err_offset = sp.err_offset;
        if (sp.seen_to < hi) {
            if (hi - lo <= SMALL_RANGE) {
                st = count_newlines(ctx, read, user, sp.seen_to, hi, &res->n_newlines);
                // (bytes BEHIND the parse error that cannot be read do not hide it: without their count nobody behind this
                // range is an anchor, as for a large range; whoever parses them for real meets the I/O failure there)
                if (st == FQH_E_IO) res->flags |= FLAG_NL_INCOMPLETE | FLAG_IO_FAILED;
                else if (st != FQH_OK) return st;
            } else {
                res->flags |= FLAG_NL_INCOMPLETE;
            }
        }
    } else {
        res->tail_len = hi - sp.end_of_records;
    }
    return FQH_OK;
}

fqh_status fqh_shard_stream_finish(fqh_ctx *ctx, fqh_read_fn read, void *user, uint64_t file_len, const uint64_t *h_all_words,
                                   int n_ranks, int rank, uint64_t slot_bytes, uint32_t n_slots, uint32_t lmax,
                                   uint64_t *d_qual_hist, uint64_t *d_base_hist, uint64_t *d_scalars, uint64_t out[2]) {
    // (ctx and read may be NULL when this rank has no gap to parse: everything else is host arithmetic on the gathered words)
    if (!h_all_words || !out || n_ranks < 1 || n_ranks > FQH_SHARD_MAX_RANKS || rank < 0 || rank >= n_ranks) return FQH_E_ARG;
    out[0] = 0;
    out[1] = FQH_NO_ERROR_KEY;
    Job job;
    classify(h_all_words, n_ranks, rank, file_len, &job);
    const uint64_t *mine = h_all_words + (size_t)rank * FQH_SHARD_STREAM_WORDS;
    if (job.failed) {
        out[1] = pack_key(job.fail_offset, rank, (int32_t)mine[W_STATUS]);
        return FQH_OK;
    }
    if (lmax && (((uint32_t)mine[W_PHASE] <= 3 && !job.own) || ((mine[W_FLAGS] & FLAG_IO_FAILED) && (uint32_t)mine[W_PHASE] > 3))) {
        // this rank streamed under a line phase that is not the true one, or up to bytes it could not read: what it counted
        // is void (the gap's parser counts its bytes under the true phase)
        if (!ctx || !d_qual_hist || !d_base_hist || !d_scalars) return FQH_E_ARG;
        fqh_status st = fqh_memset(ctx, d_qual_hist, 0, (uint64_t)lmax * 256 * sizeof(uint64_t));
        if (st == FQH_OK) st = fqh_memset(ctx, d_base_hist, 0, (uint64_t)lmax * 8 * sizeof(uint64_t));
        if (st == FQH_OK) st = fqh_memset(ctx, d_scalars, 0, FQH_NSCALARS * sizeof(uint64_t));
        if (st != FQH_OK) return st;
    }
    if (job.gap) {
        if (!ctx || !read || n_slots < 2 || (lmax && (!d_qual_hist || !d_base_hist || !d_scalars))) return FQH_E_ARG;
        Span g;
        fqh_status st = stream_span(ctx, read, nullptr, user, job.gap_from, job.gap_to, job.gap_final, slot_bytes, n_slots, lmax, d_qual_hist,
                                    d_base_hist, d_scalars, &g);
        if (st != FQH_OK) return st;
        out[0] += g.n_records;
        if (g.status != FQH_OK) {
            out[1] = pack_key(g.err_offset, rank, g.status);
            return FQH_OK;
        }
        if (!job.gap_final && g.end_of_records != job.gap_to) {
            // the anchor's phase is the true one and everything in front of it parses: its first record IS a boundary of
            // the sequential parse.  Not landing on it means the words are not what the ranks' runs produced.
            ctx->err = "fqh_shard_stream_finish: the parse of the gap in front of this rank does not end at its first record (inconsistent words)";
            return FQH_E_DEVICE;
        }
    }
    if (job.own) {
        out[0] += mine[W_RECORDS];
        if ((int32_t)mine[W_STATUS] != FQH_OK) out[1] = pack_key(mine[W_ERR_OFFSET], rank, (int32_t)mine[W_STATUS]);
    }
    return FQH_OK;
}

}  // extern "C"
