// shard_stream.hip — the byte-range sharded, host-streamed mode (BASELINE.json configs[4]) under the C ABI:
// fqh_shard_stream_run / fqh_shard_stream_finish / fqh_error_key_unpack.
//
// The reference's analogue is Parser::parallel_each (src/lib.rs:509-565): per-worker results are gathered at the end
// (src/lib.rs:553-559) and a parse error is what the whole call returns (src/lib.rs:544-547, 561-564).  Here the file is cut
// at arbitrary byte offsets, one rank = one GPU = one pinned ring; a rank cannot wait for the ranks in front of it (they
// stream for seconds), so it works PHASE-FREE and the ranks talk once, at the end:
//   1. rank r > 0 uploads a few MiB from the start of its range and asks fqh_shard_align for the line phase (newlines in
//      front of the range, mod 4) and for the offset R of its first record: the one phase under which the window parses;
//   2. it streams [lo + R, hi) through fqh_stream_* exactly like a file of its own (carry zero at lo + R; every record
//      validated in the reference's order, src/records.rs:201-247; histograms added on the way).  What is left behind its last
//      complete record is its TAIL; the bytes [lo, lo + R) are its HEAD;                         -> fqh_shard_stream_run
//   3. one all-gather of FQH_SHARD_STREAM_WORDS words + the tail bytes per rank (fqh_allgather, or the host's own collective);
//   4. every rank checks its phase against the TRUE newline count of the ranks in front of it — validity under the true line
//      phase is what the sequential parser computes — parses the STITCH = tail of rank r-1 + its own
//      head as a file of exactly one record, and packs its first error as (global record, kind) into one u64 key;
//                                                                                                 -> fqh_shard_stream_finish
//   5. one all-reduce SUM of [records, scalars, histograms] and one all-reduce MIN of the keys (fqh_allreduce_u64 /
//      fqh_allreduce_min_u64): every rank learns the first error in FILE order — the error Parser::each would have returned —
//      or the totals.
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "ctx.h"

namespace {
constexpr uint64_t ALIGN_WINDOW = 4ull << 20;

struct DevBuf {  // device scratch of the two calls (the alignment window, the stitch's little file)
    fqh_ctx *ctx;
    void *p = nullptr;
    explicit DevBuf(fqh_ctx *c) : ctx(c) {}
    ~DevBuf() {
        if (p) (void)fqh_dev_free(ctx, p);
    }
    fqh_status alloc(uint64_t bytes) { return fqh_dev_alloc(ctx, bytes, &p); }
};

// status of the packed key: FQH_E_HEADER .. FQH_E_TOO_LONG -> 0 .. 4 (the order the reference meets them inside one record,
// src/records.rs:201-247; "too long" is the Buffer's, src/lib.rs:278-283); anything else (a device error on some rank) -> 7
uint64_t pack_key(uint64_t record, int32_t status) {
    const uint64_t code = (status >= FQH_E_HEADER && status <= FQH_E_TOO_LONG) ? (uint64_t)(status - 1) : 7u;
    return (record << 3) | code;
}
}  // namespace

extern "C" {

void fqh_shard_result_words(const fqh_shard_result *r, uint64_t words[FQH_SHARD_STREAM_WORDS]) {
    if (!r || !words) return;
    words[0] = (uint64_t)(uint32_t)r->status;
    words[1] = r->n_records;
    words[2] = r->n_newlines;
    words[3] = r->phase;
    words[4] = r->head_len;
    words[5] = r->tail_len;
    words[6] = r->err_record;
    words[7] = r->err_offset;
}

fqh_status fqh_error_key_unpack(uint64_t key, int32_t *status, uint64_t *record) {
    if (!status || !record) return FQH_E_ARG;
    if (key == FQH_NO_ERROR_KEY) {
        *status = FQH_OK;
        *record = 0;
        return FQH_OK;
    }
    const uint32_t code = (uint32_t)(key & 7u);
    *status = code <= 4 ? (int32_t)code + 1 : FQH_E_DEVICE;
    *record = key >> 3;
    return FQH_OK;
}

fqh_status fqh_shard_stream_run(fqh_ctx *ctx, fqh_read_fn read, void *user, uint64_t lo, uint64_t hi, uint64_t file_len,
                                uint64_t slot_bytes, uint32_t n_slots, uint32_t lmax, uint64_t *d_qual_hist,
                                uint64_t *d_base_hist, uint64_t *d_scalars, fqh_shard_result *res, uint8_t *h_head,
                                uint64_t head_cap, uint8_t *h_tail, uint64_t tail_cap) {
    if (!ctx || !read || !res || hi < lo || hi > file_len || n_slots < 2) return FQH_E_ARG;
    const bool stats = lmax != 0;
    if (stats && (!d_qual_hist || !d_base_hist || !d_scalars)) return FQH_E_ARG;
    *res = fqh_shard_result{};
    res->status = FQH_OK;
    if (hi == lo) {  // an empty range: the rank takes part in the exchange and contributes nothing (its neighbours stitch across it)
        res->phase = FQH_SHARD_EMPTY;
        return FQH_OK;
    }
    uint64_t R = 0;
    if (lo > 0 && hi > lo) {
        // ---- where does this shard's first record begin, and at which line phase does the shard start?
        const uint64_t w = std::min<uint64_t>(ALIGN_WINDOW, hi - lo);
        std::vector<uint8_t> hostw(w + 1);
        if (read(user, hostw.data(), lo - 1, w + 1) != 0) {  // one byte more in front: is it a newline?
            ctx->err = "fqh_shard_stream_run: the read callback failed";
            return FQH_E_IO;
        }
        const bool prev_nl = hostw[0] == '\n';
        bool any_line_start = false;  // (a line starts at lo + i iff the byte in front of it is a newline)
        for (uint64_t i = 0; i < w && !any_line_start; ++i) any_line_start = hostw[i] == '\n';
        if (!any_line_start && w == hi - lo) {
            // not one line starts inside the range, so no record does: it lies inside one line of one record, and there is no
            // line phase to settle — the record in progress runs through this rank (FQH_SHARD_PASS, below)
            if (w > tail_cap || !h_tail) {
                ctx->err = "fqh_shard_stream_run: tail_cap is smaller than a byte range that holds no record start";
                return FQH_E_CAPACITY;
            }
            memcpy(h_tail, hostw.data() + 1, w);
            res->tail_len = w;
            res->phase = FQH_SHARD_PASS;
            for (uint64_t i = 0; i < w; ++i) res->n_newlines += hostw[1 + i] == '\n';
            return FQH_OK;
        }
        uint32_t phase = 0;
        auto align_on = [&](uint64_t wn) -> fqh_status {  // line phase and first record start from the wn bytes at lo
            DevBuf win(ctx);
            fqh_status e = win.alloc(wn + 16);
            if (e != FQH_OK) return e;
            e = fqh_memcpy_h2d(ctx, win.p, hostw.data() + 1, wn);
            if (e != FQH_OK) return e;
            return fqh_shard_align(ctx, (const uint8_t *)win.p, wn, prev_nl ? 1 : 0, &phase, &R);
        };
        // The window only settles the phase and finds the first record start, so it may reach BEHIND the range: a few lines
        // settle nothing — worse, the phase that "gets furthest" in them can be the wrong one (a quality line that starts with
        // '@' right behind the cut looks like a header: tools/fuzz_sharded.py with cuts close together) — while up to 4 MiB of
        // the file behind lo do, whatever the range's own size.  The first record start may then lie behind the range: it holds
        // none (FQH_SHARD_PASS below).  If the long window holds a parse error and no phase stands out, the error may lie in a
        // later rank's bytes: the range's own bytes decide then, as before.
        const uint64_t w2 = std::min<uint64_t>(ALIGN_WINDOW, file_len - lo);
        fqh_status st = FQH_E_ARG;
        if (w2 > w) {
            hostw.resize(w2 + 1);
            if (read(user, hostw.data(), lo - 1, w2 + 1) != 0) {
                ctx->err = "fqh_shard_stream_run: the read callback failed";
                return FQH_E_IO;
            }
            st = align_on(w2);
            if (st != FQH_OK && st != FQH_E_ARG && st != FQH_E_HEADER) return st;
        }
        if (st != FQH_OK) st = align_on(w);
        if (st == FQH_E_ARG || (st == FQH_E_HEADER && hi - lo < FQH_BUFSIZE)) {
            // several line phases validate, or none does in a range that need not even hold one record start (the reference
            // accepts records of up to BUFSIZE bytes): too few lines to tell, and a parse error could not be told from "too
            // little to see".  Not a property of the file: the caller has cut it too finely (merge the range with a neighbour;
            // an EMPTY range is fine)
            ctx->err = "fqh_shard_stream_run: the byte range is too small to settle its line phase (it must hold a few records)";
            return FQH_E_ARG;
        }
        if (st == FQH_E_HEADER) {
            // the window holds a parse error: reported as this shard's error at its start
            res->status = FQH_E_HEADER;
            res->err_offset = lo;
            return FQH_OK;
        }
        if (st != FQH_OK) return st;
        res->phase = phase;
        if (R >= hi - lo) {
            // no record starts inside the range (it lies inside one record, or ends exactly where the next one begins): the
            // record in progress runs THROUGH this rank.  All of its bytes are handed on as its tail; the stitch in front of the
            // next rank that holds a record start (or the end of the file) is parsed across it.  The phase a few bytes settle on
            // means nothing and is not looked at.
            const uint64_t n = hi - lo;
            if (n > tail_cap || !h_tail) {
                ctx->err = "fqh_shard_stream_run: tail_cap is smaller than a byte range that holds no record start";
                return FQH_E_CAPACITY;
            }
            memcpy(h_tail, hostw.data() + 1, n);
            res->tail_len = n;
            res->phase = FQH_SHARD_PASS;
            for (uint64_t i = 0; i < n; ++i) res->n_newlines += hostw[1 + i] == '\n';
            return FQH_OK;
        }
        if (R > head_cap || (R && !h_head)) {
            ctx->err = "fqh_shard_stream_run: head_cap is smaller than the shard's head";
            return FQH_E_CAPACITY;
        }
        if (R) memcpy(h_head, hostw.data() + 1, R);
        res->head_len = R;
        for (uint64_t i = 0; i < R; ++i) res->n_newlines += hostw[1 + i] == '\n';
    }
    uint64_t pos = lo + R;
    if (pos >= hi) return FQH_OK;  // nothing but the head
    fqh_stream *sp = nullptr;
    fqh_status st = fqh_stream_create(ctx, slot_bytes, n_slots, stats ? FQH_STREAM_STATS : 0u, &sp);
    if (st != FQH_OK) return st;
    struct Closer {
        fqh_stream *s;
        ~Closer() { fqh_stream_destroy(s); }
    } closer{sp};
    if (stats) {
        st = fqh_stream_set_stats(sp, lmax, d_qual_hist, d_base_hist, d_scalars);
        if (st != FQH_OK) return st;
    }
    const bool is_last_shard = hi >= file_len;
    bool done_reading = false;
    uint64_t submitted = 0, collected = 0;
    for (;;) {
        while (!done_reading) {
            uint8_t *dst = nullptr;
            uint64_t cap = 0;
            st = fqh_stream_acquire(sp, &dst, &cap);
            if (st == FQH_E_CAPACITY) break;  // the ring is full: collect first
            if (st != FQH_OK) return st;
            const uint64_t n = std::min<uint64_t>(cap, hi - pos);
            if (read(user, dst, pos, n) != 0) {
                ctx->err = "fqh_shard_stream_run: the read callback failed";
                return FQH_E_IO;
            }
            pos += n;
            done_reading = pos >= hi;
            st = fqh_stream_submit(sp, n, (done_reading && is_last_shard) ? 1 : 0);
            if (st != FQH_OK) return st;
            ++submitted;
        }
        if (collected == submitted) break;
        fqh_chunk c;
        st = fqh_stream_collect(sp, &c);
        if (st != FQH_OK) return st;
        ++collected;
        res->n_records += c.n_records;
        const uint64_t end_of_records = lo + R + c.h_rec_start[c.n_records];  // (the stream's file offsets count from lo + R)
        if (c.parse_status != FQH_OK) {
            res->status = c.parse_status;
            res->err_record = c.err_record;
            res->err_offset = lo + R + c.err_offset;
            (void)fqh_stream_release(sp);
            break;
        }
        if (collected == submitted && done_reading) {
            // what is left behind the last complete record: in pinned memory, in front of / inside the last chunk
            const uint64_t tail_len = hi - end_of_records;
            if (tail_len > tail_cap || (tail_len && !h_tail)) {
                (void)fqh_stream_release(sp);
                ctx->err = "fqh_shard_stream_run: tail_cap is smaller than the shard's tail";
                return FQH_E_CAPACITY;
            }
            const int64_t off = (int64_t)(end_of_records - (lo + R)) - (int64_t)c.base_offset;  // relative to h_data (may be negative: in the lead)
            if (tail_len) memcpy(h_tail, c.h_data + off, tail_len);
            res->tail_len = tail_len;
        }
        st = fqh_stream_release(sp);
        if (st != FQH_OK) return st;
    }
    fqh_carry cy;
    st = fqh_stream_carry(sp, &cy);
    if (st != FQH_OK) return st;
    res->n_newlines += cy.nl_count;
    return FQH_OK;
}

fqh_status fqh_shard_stream_finish(fqh_ctx *ctx, const uint64_t *h_all_words, const uint8_t *h_all_tails, uint64_t tail_stride,
                                   int n_ranks, int rank, const uint8_t *h_head, uint32_t lmax, uint64_t *d_qual_hist,
                                   uint64_t *d_base_hist, uint64_t *d_scalars, uint64_t out[2]) {
    // (ctx may be NULL when this rank has no stitch to parse — both the previous rank's tail and its own head are empty: the
    // phase check and the key are host arithmetic on the gathered words)
    if (!h_all_words || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks) return FQH_E_ARG;
    const uint64_t *mine = h_all_words + (size_t)rank * FQH_SHARD_STREAM_WORDS;
    auto W = [&](int j) { return h_all_words + (size_t)j * FQH_SHARD_STREAM_WORDS; };
    auto empty = [&](int j) { return W(j)[3] == FQH_SHARD_EMPTY; };
    auto pass = [&](int j) { return W(j)[3] == FQH_SHARD_PASS; };   // bytes, but no record start: the stitch runs across it
    auto prev_of = [&](int j) {  // the nearest rank in front of j that holds a record start (-1: none)
        int p = j - 1;
        while (p >= 0 && (empty(p) || pass(p))) --p;
        return p;
    };
    auto carried = [&](int p, int j) {  // bytes that reach rank j from the ranks in front of it: rank p's tail + every rank passed through
        uint64_t n = p >= 0 ? W(p)[5] : 0;
        for (int q = p + 1; q < j; ++q)
            if (pass(q)) n += W(q)[5];
        return n;
    };
    // ---- what lies in front of this rank: true newline count, records (the streamed ones and one per non-empty stitch)
    uint64_t nl_before = 0, rec_before = 0;
    bool earlier_error = false;  // a rank in front of this one stopped at an error of its own (or parsed under a wrong phase): its
                                 // key is the smaller one in file order, and what this rank derives from its counts is not to be used
    for (int j = 0; j < rank; ++j) {
        if (empty(j)) continue;
        const uint64_t *w = W(j);
        if (pass(j)) {  // (its bytes belong to the stitch of a later rank)
            nl_before += w[2];
            continue;
        }
        const int p = prev_of(j);
        if (p >= 0 && (carried(p, j) + w[4]) != 0) ++rec_before;  // rank j's stitch
        if ((int32_t)w[0] != FQH_OK || (p >= 0 && (nl_before & 3) != w[3])) earlier_error = true;
        nl_before += w[2];
        rec_before += w[1];
    }
    uint64_t records = 0;
    uint64_t key = FQH_NO_ERROR_KEY;
    if (empty(rank)) {
        out[0] = 0;
        out[1] = FQH_NO_ERROR_KEY;
        return FQH_OK;
    }
    bool last_bytes = true;  // no rank behind this one holds bytes
    for (int j = rank + 1; j < n_ranks; ++j)
        if (!empty(j)) last_bytes = false;
    const bool through = pass(rank);
    if (through && !last_bytes) {  // the record in progress ends in a later rank: that one parses the stitch
        out[0] = 0;
        out[1] = FQH_NO_ERROR_KEY;
        return FQH_OK;
    }
    const int prev = prev_of(rank);
    // ---- the record that straddles the cut(s) in front of this rank: tail of the last rank with a record start + the ranks it
    // runs through + own head, a file of its own.  (A rank WITHOUT a record start at the end of the file parses what has reached
    // it, itself included, as the file's end: a last record without its newline, or a truncated one.)
    if (prev >= 0) {
        const uint64_t tl = carried(prev, rank) + (through ? mine[5] : 0), hl = through ? 0 : mine[4];
        if (tl + hl) {
            if (!ctx || (tl && !h_all_tails) || (hl && !h_head)) return FQH_E_ARG;
            std::vector<uint8_t> file(tl + hl);
            uint64_t at = 0;
            for (int q = prev; q <= rank; ++q) {
                if (q != prev && !pass(q)) continue;
                if (q == rank && !through) continue;
                const uint64_t n = W(q)[5];
                if (n > tail_stride) return FQH_E_ARG;
                if (n) memcpy(file.data() + at, h_all_tails + (size_t)q * tail_stride, n);
                at += n;
            }
            if (hl) memcpy(file.data() + at, h_head, hl);
            DevBuf d(ctx);
            fqh_status st = d.alloc(tl + hl + 16);
            if (st != FQH_OK) return st;
            st = fqh_memcpy_h2d(ctx, d.p, file.data(), tl + hl);
            if (st != FQH_OK) return st;
            fqh_summary s = {};
            if (lmax) st = fqh_stats(ctx, (const uint8_t *)d.p, tl + hl, 1, nullptr, lmax, d_qual_hist, d_base_hist, d_scalars, &s, nullptr);
            else st = fqh_scan(ctx, (const uint8_t *)d.p, tl + hl, 1, nullptr, nullptr, 0, &s, nullptr);
            if (st != FQH_OK) return st;
            if (s.parse_status != FQH_OK) key = pack_key(rec_before + s.n_records, s.parse_status);
            else if (s.n_records != 1) key = pack_key(rec_before + s.n_records, FQH_E_TRUNCATED);  // (a tail + head of one record: cannot happen)
            records += s.n_records;
        }
    }
    if (through) {  // (nothing of its own behind the stitch)
        out[0] = records;
        out[1] = earlier_error ? FQH_NO_ERROR_KEY : key;
        return FQH_OK;
    }
    // ---- the phase this rank parsed under against the true one
    if (key == FQH_NO_ERROR_KEY && prev >= 0 && (int32_t)mine[0] == FQH_OK && (nl_before & 3) != mine[3])
        key = pack_key(rec_before + records, FQH_E_HEADER);
    // ---- the rank's own records and its own first error
    if (key == FQH_NO_ERROR_KEY) {
        records += mine[1];
        if ((int32_t)mine[0] != FQH_OK) key = pack_key(rec_before + records, (int32_t)mine[0]);
    }
    out[0] = records;
    out[1] = earlier_error ? FQH_NO_ERROR_KEY : key;
    return FQH_OK;
}

}  // extern "C"
