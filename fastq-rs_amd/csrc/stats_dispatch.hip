// stats_dispatch.hip — libfastq_hip.so, host side of the statistics calls (fqh_stats*, fqh_scan_stats*, fqh_len_hist, the
// filter calls): which route a call takes — the single pass (k_scan_stats) or the exact scan followed by the histogram
// kernels — and what the context remembers about its inputs to decide that.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "ctx.h"
#include "dispatch.h"


// The single-pass route (k_scan_stats) counts every sequence / quality line that closes inside the buffer, minus the lines of
// the record in progress at its start; k_stats_edge adds that record (if the caller's buffer holds its beginning) and takes
// out the sequence line of the partial record at the end of a chunk that is not the file's last.  So whole files AND chunks
// take it, as long as the histogram fits the kernel's LDS rows and the context is not backing off from the fast path.  Anything
// else takes the two-pass route (exact index + k_stats_oct), which knows about record limits.
// What the context knows about the length of the reads it is given (fqh_ctx::rows_hint; scan_stats_rows says what it is for).
// Nothing yet: a look at the input's first 64 KiB (one small kernel and a wait of some tens of microseconds, once per context).
// Afterwards the calls themselves say: a pass that met lines beyond its rows (listed, or given up) takes the scan's longest
// record as the new bound; a pass that met none lets the bound come down to it.
constexpr uint32_t FZ_ROWS_MAX = 511;   // the most rows the single pass keeps (fused_kernels.hip: FZ_LC_MAX)
static fqh_status ensure_rows_hint(fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, const fqh_carry *in, uint32_t lmax) {
    // (A context that knows something looks again only at inputs of a GiB and more —
    // a look is some tens of microseconds, a pass over the wrong rows a whole read of the input — and only if the answer could
    // change something: rows below lmax that might be too few, or a belief in kilobase reads that keeps the pass away)
    if (!ctx->fused_enabled || !lmax || !len || !d_buf) return FQH_OK;
    // What the context believes about the reads' length, and its back-off from the single pass, belong to ONE input: the same
    // buffer again (a benchmark's steps, a resident file counted twice) or the next chunk of the same file (ring slots, a host's
    // own chunking: the carry's file offset continues).  Anything else is another input — a 300 bp file behind a 100 bp one —
    // and starts from a look of its own, with no back-off it has not earned.
    {
        const uint64_t base = in ? in->base_offset : 0;
        const bool same = ctx->hint_valid && ctx->hint_buf == d_buf && ctx->hint_len == len && ctx->hint_base == base;
        const bool next = ctx->hint_valid && base != 0 && base == ctx->hint_base + ctx->hint_len;
        if (!same && !next) {
            ctx->rows_hint = 0;
            ctx->lines_long = false;
            ctx->fused_skip = ctx->fused_backoff = 0;
        }
        ctx->hint_buf = d_buf;
        ctx->hint_len = len;
        ctx->hint_base = base;
        ctx->hint_valid = true;
    }
    if (ctx->rows_hint && (len < (1ull << 30) || (!ctx->lines_long && scan_stats_rows(lmax, ctx->rows_hint) >= std::min(lmax, FZ_ROWS_MAX)))) return FQH_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    unsigned long long peek[12] = {};
    const uint32_t nw = peek_windows(len);
    launch_peek_lines(ctx->stream, d_buf, len, in ? (uint32_t)(in->nl_count & 3u) : 0u, (unsigned long long *)ctx->d_misc);
    HIPCHK(ctx, hipMemcpyAsync(peek, ctx->d_misc, nw * 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    unsigned long long longest = 0, lines = 0, longs = 0;   // over the windows (k_peek_lines: the input's first bytes and three more)
    for (uint32_t w = 0; w < nw; ++w) {
        longest = std::max(longest, peek[3 * w]);
        lines += peek[3 * w + 1];
        longs += peek[3 * w + 2];
    }
    const uint32_t seen = (uint32_t)std::min<unsigned long long>(std::max<unsigned long long>(longest, 1), 0x7FFFFFFFull);
    // the rows go up with what a look finds, down with what the scans find (resolve()); whether MOST lines are too long for the
    // pass is what the look says (64 KiB of kilobase reads: every other line)
    if (seen > ctx->rows_hint || !ctx->rows_hint) ctx->rows_hint = seen;
    ctx->lines_long = longs * 4 > lines || (lines == 0 && len >= 65536);   // (no newline in 64 KiB)
    return FQH_OK;
}
static void update_rows_hint(fqh_ctx *ctx) {   // after a single pass whose scan stands (fused_finish)
    const uint64_t bound = ctx->last_summary.max_record_len / 2;   // no sequence / quality line of a delivered record is longer
    if (!bound) return;
    const bool beyond = ctx->h_out->stats_declined != 0 || ctx->h_out->decl_lines != 0;
    if (beyond && bound > ctx->rows_hint) ctx->rows_hint = (uint32_t)std::min<uint64_t>(bound, 0x7FFFFFFFull);   // (down: resolve())
    if (ctx->h_out->stats_declined != 0 && bound > 2 * 511) ctx->lines_long = true;   // (given up, and over records that hold such lines)
}
// A statistics call that does not take the single pass — it is backing off, or the fast path is — counts both back-offs down: the
// scan it runs instead has the fast path switched off (it needs complete line lists) and so never reaches the count-down in
// do_scan_launch.  (Until round 5 a context whose fast path had failed once — one file of kilobase reads — and which was then given
// nothing but fqh_stats calls never tried the fast path, or the single pass, again.)  Not for the second pass of a call that has
// just given its single pass up (fused_enabled is off for that one).
static void count_down_backoffs(fqh_ctx *ctx) {
    if (!ctx->fused_enabled) return;
    if (ctx->fused_skip) --ctx->fused_skip;
    if (ctx->spec_enabled && ctx->spec_skip) --ctx->spec_skip;
}
static bool fused_eligible(const fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, int is_final, const fqh_carry *in,
                           uint32_t lmax, uint64_t lead_len, uint64_t n_limit) {
    if (!ctx->fused_enabled || !ctx->spec_enabled || ctx->exact_holds || ctx->spec_skip || ctx->list_cap != LIST_CAP_DEFAULT) return false;
    if (ctx->fused_skip) return false;   // backing off after a pass that was given up (fused_finish)
    // (chunks with a carry, chunks that are not the file's last and lead bytes are fine: k_stats_edge settles the records at
    // the chunk's two ends; a record LIMIT is not — the kernel counts every line it meets — except for the streaming ring,
    // which commits only after it knows that the limit does not bite: f_defer_commit)
    if ((n_limit != UINT64_MAX && !ctx->f_defer_commit) || !scan_stats_supports(lmax, ctx->rows_hint, ctx->lines_long) || !len) return false;
    if (in && in->back[in->nl_count & 3] > in->base_offset) return false;
    (void)lead_len;
    (void)is_final;
    if (same_scan(ctx, d_buf, len, is_final, in) && ctx->index_full) return false;  // a full index is there: second pass only
    return true;
}
static fqh_status fused_launch(fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, int is_final, const fqh_carry *in,
                               uint64_t *d_rec_start, uint64_t cap, uint32_t lmax, uint64_t *d_qual_hist,
                               uint64_t *d_base_hist, uint64_t *d_scalars, uint64_t lead_len = 0) {
    ctx->fused = true;
    ctx->f_lead = lead_len;
    ctx->f_lmax = lmax;
    ctx->f_rows = scan_stats_rows(lmax, ctx->rows_hint);
    ctx->f_qual = d_qual_hist;
    ctx->f_base = d_base_hist;
    ctx->f_scalars = d_scalars;
    fqh_status st = do_scan_launch(ctx, d_buf, len, is_final, in, d_rec_start, cap);
    if (st != FQH_OK) ctx->fused = false;
    return st;
}
// Finish of a single-pass launch: the scan's finish (which reruns the exact path if the fast path's proof failed);
// *two_pass = the histograms were NOT committed and have to be counted over the exact index.
static fqh_status fused_finish(fqh_ctx *ctx, fqh_summary *out, fqh_carry *carry_out, bool *two_pass) {
    fqh_status st = do_scan_finish(ctx, out, carry_out);
    ctx->fused = false;
    // (stats_declined: the kernel met lines it can neither count nor hand to k_stats_declined — kilobase reads, more dirty
    // batches than the dump area holds.  The scan's result stands, on the fast path; only the histograms take a second pass,
    // and the fast path's back-off does not hear of it: nothing was wrong with the parse)
    *two_pass = !ctx->used_spec || ctx->h_out->stats_declined != 0;
    // A pass that was given up cost a whole read of the input for nothing, and the next chunk of the same file will do the same:
    // the context's next 1, 2, 4 .. 64 statistics calls go straight to the two-pass route (fqh_stats_launch counts them down), a
    // pass that commits resets the count — the fast path's own rule (do_scan_finish), for the same reason.
    if (ctx->used_spec) update_rows_hint(ctx);
    if (ctx->used_spec && ctx->h_out->stats_declined != 0) {
        ctx->fused_backoff = ctx->fused_backoff ? (ctx->fused_backoff < 64 ? ctx->fused_backoff * 2 : 64) : 1;
        ctx->fused_skip = ctx->fused_backoff;
    } else if (ctx->used_spec) {
        ctx->fused_backoff = 0;
    }
    ctx->stats_route = *two_pass ? 0 : (ctx->h_out->decl_batches || ctx->h_out->decl_lines) ? 2 : 1;
    if (ctx->used_spec) ctx->timing.stats_ms = ctx->timing.index_ms;  // the one kernel that read the input
    return st;
}

// The streaming ring's entry to the single pass: scan + histograms of one slot, the commit kernels held back until the ring
// has replayed the reference's Buffer over the slot's boundaries (a "record too long" in the middle of a slot ends the stream
// there: the records behind it must not count).  *fused = false: not eligible, nothing was launched.
fqh_status fqh_internal_fused_launch(fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, int is_final, const fqh_carry *in,
                                     uint64_t *d_rec_start, uint64_t cap, uint32_t lmax, uint64_t *d_qual_hist,
                                     uint64_t *d_base_hist, uint64_t *d_scalars, uint64_t lead_len, bool *fused) {
    *fused = false;
    if (!ctx || ctx->pending || ctx->stats_pending) return FQH_E_ARG;
    if (fqh_status hs = ensure_rows_hint(ctx, d_buf, len, in, lmax); hs != FQH_OK) return hs;
    ctx->f_defer_commit = true;
    const bool ok = fused_eligible(ctx, d_buf, len, is_final, in, lmax, lead_len, 0);
    if (!ok) {
        count_down_backoffs(ctx);
        ctx->f_defer_commit = false;
        return FQH_OK;
    }
    ctx->last_valid = false;
    fqh_status st = fused_launch(ctx, d_buf, len, is_final, in, d_rec_start, cap, lmax, d_qual_hist, d_base_hist, d_scalars, lead_len);
    // (f_defer_commit stays set until the launch's finish: the finish may run the fast path a second time — a context that
    // meets its first tile with more record starts than two lines hold allocates the list area and reruns — and that run's
    // commit must be held back as well)
    if (st == FQH_OK) *fused = true;
    else ctx->f_defer_commit = false;
    return st;
}
// after fqh_internal_scan_finish of such a launch: did the single pass stand (a commit is owed), and enqueue it
bool fqh_internal_fused_owed(const fqh_ctx *ctx) { return ctx->f_commit_owed && ctx->used_spec && !ctx->h_out->stats_declined; }
void fqh_internal_fused_commit(fqh_ctx *ctx) {
    if (ctx->f_commit_owed && ctx->used_spec && !ctx->h_out->stats_declined) enqueue_fused_commit(ctx);
    ctx->f_commit_owed = false;
}
void fqh_internal_fused_drop(fqh_ctx *ctx) { ctx->f_commit_owed = false; }

// lead_len: bytes in front of d_buf that are valid device memory and hold the beginning of the
// record in progress at the chunk start; n_limit: count at most this many records of the chunk.
fqh_status fqh_internal_stats_launch(fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, int is_final,
                                     const fqh_carry *in, uint32_t lmax, uint64_t *d_qual_hist,
                                     uint64_t *d_base_hist, uint64_t *d_scalars, uint64_t lead_len,
                                     uint64_t n_limit) {
    if (!ctx) return FQH_E_ARG;
    if (!d_qual_hist || !d_base_hist || !d_scalars || lmax == 0) return fail(ctx, FQH_E_ARG, "NULL histogram or lmax == 0");
    if (ctx->pending || ctx->stats_pending) return fail(ctx, FQH_E_ARG, "a launch is already pending");
    fqh_status st;
    if (st = ensure_rows_hint(ctx, d_buf, len, in, lmax); st != FQH_OK) return st;
    if (fused_eligible(ctx, d_buf, len, is_final, in, lmax, lead_len, n_limit)) {
        // one read of the input: scan + histograms in k_scan_stats (src/lib.rs:226-237 hands each record to the
        // closure that reads seq()/qual(): one pass).  fqh_stats_finish falls back to the two-pass route if
        // the fast path's proof fails.
        st = fused_launch(ctx, d_buf, len, is_final, in, nullptr, 0, lmax, d_qual_hist, d_base_hist, d_scalars, lead_len);
        if (st != FQH_OK) return st;
        ctx->stats_pending = true;
        return FQH_OK;
    }
    count_down_backoffs(ctx);
    if (!same_scan(ctx, d_buf, len, is_final, in)) {
        // the histogram kernel needs complete line lists: scan on the exact path right away instead of
        // taking the fast path and indexing a second time
        const bool spec = ctx->spec_enabled;
        ctx->spec_enabled = false;
        if (lmax > 256) {
            // rows beyond the single pass: the caller expects reads of more than 256 columns, which k_stats_long counts over the
            // record index — the scan's emit step writes it on the way (one entry per 512 bytes of input fits; denser input, or
            // reads that turn out short, take the separate emit below as before)
            const uint64_t cap = len / 512 + 16;
            if (ctx->idx_cap < cap) {
                (void)hipFree(ctx->idx);
                ctx->idx = nullptr;
                ctx->idx_cap = 0;
                HIPCHK(ctx, hipMalloc((void **)&ctx->idx, cap * sizeof(fqh_idx_record)));
                ctx->idx_cap = cap;
            }
            ctx->scan_idx = ctx->idx;
            ctx->scan_idx_cap = ctx->idx_cap;
        }
        st = do_scan_launch(ctx, d_buf, len, is_final, in, nullptr, 0);
        ctx->scan_idx = nullptr;
        if (st == FQH_OK) st = do_scan_finish(ctx, nullptr, nullptr);
        ctx->spec_enabled = spec;
        if (st != FQH_OK) return st;
    } else {
        ctx->idx_emitted = false;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    st = ensure_full_index(ctx);  // the histogram kernel walks complete line lists
    if (st != FQH_OK) return st;
    const fqh_timing scan_t = ctx->timing;
    const uint64_t n = std::min<uint64_t>(ctx->last_summary.n_records, n_limit);
    // the record in progress at the chunk start began in an earlier chunk: the line lists do not
    // cover it; it is counted separately (k_stats_head) when the caller's buffer holds its beginning
    const uint64_t back0 = ctx->carry_in.back[ctx->carry_in.nl_count & 3];
    const uint64_t skip = (n && back0 > 0) ? 1 : 0;
    const bool head = skip && lead_len >= back0;
    hipStream_t s = ctx->stream;
    HIPCHK(ctx, hipEventRecord(ctx->ev[4], s));
    HIPCHK(ctx, hipEventRecord(ctx->ev[5], s));
    if (n > skip) {
        // reads longer than the kernel's 256 LDS rows are counted in several passes, which share one bit per record
        // and alphabet flag (behind the partial histograms in the scratch)
        const uint32_t max_line = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(ctx->last_summary.max_record_len / 2, len + ctx->carry_in.back[3]), 0xFFFFFFFFu);
        // (by the READS' length, not by the caller's rows: lmax is the caller's choice — the first 150 cycles of 5 kbp reads — and
        // what lies beyond it is still looked at, for the alphabet flags: k_stats_oct does that byte by byte, 10.9 ms per 4 GiB of
        // 5 kbp reads with lmax = 150; k_stats_long's column blocks beyond lmax only look, at the speed of their loads, 1.3 ms)
        const bool passes = (max_line ? max_line : lmax) > 256;
        // Reads beyond the 256 rows: one walk over the record index (k_stats_long; its blocks' rows take the place of k_stats_oct's
        // partial histograms in the scratch).  Round 3 kept two passes of k_stats_oct for reads of up to 512 columns (2340 against
        // 1660 GB/s at 300 bp); since k_stats_long runs in one round of blocks the two are level at 300 bp (1.91 / 1.97 ms per
        // 4 GiB), k_stats_long is ahead from there (500 bp: 1.98 / 1.34) and it alone keeps 128 quality bins in LDS: with
        // qualities beyond '`' the passes of k_stats_oct count byte by byte in device memory (300 bp, 80 % '~': 146 ms).
        const bool long_route = passes;
        const size_t hist_bytes = long_route ? std::max(stats_oct_scratch_bytes(lmax, ctx->n_cu), stats_long_scratch_bytes(n - skip, len, max_line, ctx->n_cu))
                                             : stats_oct_scratch_bytes(lmax, ctx->n_cu);
        const uint64_t flag_words = passes ? (n - skip + 31) / 32 + 1 : 0;
        const size_t need = hist_bytes + ((size_t)flag_words * 2 + 1) * sizeof(uint32_t);
        if (need > ctx->stats_scratch_bytes) {
            (void)hipFree(ctx->stats_scratch);
            ctx->stats_scratch = nullptr;
            ctx->stats_scratch_bytes = 0;
            HIPCHK(ctx, hipMalloc((void **)&ctx->stats_scratch, need));
            ctx->stats_scratch_bytes = need;
        }
        const uint64_t r0 = ctx->carry_in.nl_count >> 2;
        StatsArgs sa = {};
        sa.buf = d_buf;
        sa.len = len;
        sa.valid_end = ctx->last_summary.bytes_consumed;
        sa.nl_count = ctx->carry_in.nl_count;
        sa.line_lo = 4 * (r0 + skip);
        sa.line_hi = 4 * (r0 + n);
        sa.list = ctx->list;
        sa.list_cap = ctx->list_cap;
        sa.tile_count = ctx->tile_count;
        sa.tile_prefix = ctx->tile_prefix;
        sa.block_prefix = ctx->block_prefix;
        sa.n_tiles = ctx->args.n_tiles;
        sa.lmax = lmax;
        // a delivered record holds its sequence and its quality line, of one length: neither is longer than half of it
        sa.max_line = max_line;
        sa.scratch = ctx->stats_scratch;
        if (passes) {
            sa.flagmap = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(ctx->stats_scratch) + hist_bytes);
            sa.flag_words = flag_words;
            sa.cr_flag = sa.flagmap + flag_words * 2;
            HIPCHK(ctx, hipMemsetAsync(sa.flagmap, 0, ((size_t)flag_words * 2 + 1) * sizeof(uint32_t), s));
        }
        sa.qual_hist = (unsigned long long *)d_qual_hist;
        sa.base_hist = (unsigned long long *)d_base_hist;
        sa.scalars = (unsigned long long *)d_scalars;
        if (long_route) {
            // did the scan's own emit step write the whole index?  (Decided BEFORE the array may be replaced: a free followed
            // by an allocation hands the same address out again, with the old entries in it and nothing behind them.)
            const bool have_idx = ctx->idx_emitted && ctx->used_spec == false && ctx->args.idx == ctx->idx && n <= ctx->args.idx_cap;
            if (!have_idx) {
                if (ctx->idx_cap < n) {
                    (void)hipFree(ctx->idx);
                    ctx->idx = nullptr;
                    ctx->idx_cap = 0;
                    HIPCHK(ctx, hipMalloc((void **)&ctx->idx, n * sizeof(fqh_idx_record)));
                    ctx->idx_cap = n;
                }
                st = emit_index(ctx, ctx->idx, n);
                if (st != FQH_OK) return st;
            }
            HIPCHK(ctx, launch_stats_long(s, d_buf, len, ctx->carry_in.base_offset, ctx->idx + skip, n - skip, lmax, max_line, sa.flagmap,
                                          flag_words, sa.qual_hist, sa.base_hist, sa.scalars, ctx->n_cu, ctx->stats_scratch));
        } else {
            HIPCHK(ctx, launch_stats_oct(s, sa, ctx->n_cu));
        }
    }
    if (head) {
        StatsArgs sa = {};
        sa.buf = d_buf;
        sa.len = len;
        sa.nl_count = ctx->carry_in.nl_count;
        sa.list = ctx->list;
        sa.list_cap = ctx->list_cap;
        sa.tile_count = ctx->tile_count;
        sa.n_tiles = ctx->args.n_tiles;
        sa.lmax = lmax;
        sa.qual_hist = (unsigned long long *)d_qual_hist;
        sa.base_hist = (unsigned long long *)d_base_hist;
        sa.scalars = (unsigned long long *)d_scalars;
        launch_stats_head(s, sa, ctx->carry_in.back);
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev[6], s));
    HIPCHK(ctx, hipGetLastError());
    ctx->timing = scan_t;
    ctx->stats_pending = true;
    return FQH_OK;
}

extern "C" {

fqh_status fqh_stats_launch(fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, int is_final,
                            const fqh_carry *in, uint32_t lmax, uint64_t *d_qual_hist,
                            uint64_t *d_base_hist, uint64_t *d_scalars) {
    return fqh_internal_stats_launch(ctx, d_buf, len, is_final, in, lmax, d_qual_hist, d_base_hist, d_scalars, 0,
                                     UINT64_MAX);
}

fqh_status fqh_stats_launch_lead(fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, uint64_t lead_len, int is_final,
                                 const fqh_carry *in, uint32_t lmax, uint64_t *d_qual_hist,
                                 uint64_t *d_base_hist, uint64_t *d_scalars) {
    return fqh_internal_stats_launch(ctx, d_buf, len, is_final, in, lmax, d_qual_hist, d_base_hist, d_scalars,
                                     lead_len, UINT64_MAX);
}

fqh_status fqh_stats_finish(fqh_ctx *ctx, fqh_summary *out, fqh_carry *carry_out) {
    if (!ctx) return FQH_E_ARG;
    if (!ctx->stats_pending) return fail(ctx, FQH_E_ARG, "no stats pending");
    ctx->stats_pending = false;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    fqh_status cap_st = ctx->stats_cap_st;  // the scan in front of the histograms found d_rec_start too short
    ctx->stats_cap_st = FQH_OK;
    if (!ctx->fused) ctx->stats_route = 0;
    if (ctx->fused) {
        const uint8_t *buf = ctx->args.buf;
        const uint64_t len = ctx->args.len;
        const int is_final = ctx->args.is_final;
        const fqh_carry cin = ctx->carry_in;
        const uint32_t lmax = ctx->f_lmax;
        const uint64_t lead = ctx->f_lead;
        uint64_t *qh = ctx->f_qual, *bh = ctx->f_base, *sc = ctx->f_scalars;
        bool two_pass = false;
        const fqh_status scan_st = fused_finish(ctx, out, carry_out, &two_pass);
        if (scan_st != FQH_OK && scan_st != FQH_E_CAPACITY) return scan_st;
        // (FQH_E_CAPACITY: the histograms are complete and the summary exact, d_rec_start was too short — both routes
        // report it, like fqh_scan)
        if (!two_pass) return scan_st;
        // the exact path has rerun the scan (same buffer, full index), or the single pass kept the scan and declined the
        // count: the histogram kernels count over the full index of these very bytes (built now, in the second case)
        ctx->trust_index = true;   // (the scan of these very bytes has just finished inside this call)
        const bool fe = ctx->fused_enabled;
        ctx->fused_enabled = false;  // (not the single pass again)
        fqh_status st = fqh_internal_stats_launch(ctx, buf, len, is_final, &cin, lmax, qh, bh, sc, lead, UINT64_MAX);
        ctx->fused_enabled = fe;
        ctx->trust_index = false;
        if (st != FQH_OK) return st;
        ctx->stats_pending = false;
        cap_st = scan_st;
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    float ms = 0;
    if (hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[5]) == hipSuccess) ctx->timing.emit_ms = ms;
    if (hipEventElapsedTime(&ms, ctx->ev[5], ctx->ev[6]) == hipSuccess) ctx->timing.stats_ms = ms;
    if (hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[6]) == hipSuccess) ctx->timing.total_ms = ms;
    if (out) *out = ctx->last_summary;
    if (carry_out) *carry_out = ctx->last_carry_out;
    if (cap_st != FQH_OK) return fail(ctx, cap_st, "d_rec_start capacity < n_records + 1");
    return FQH_OK;
}

fqh_status fqh_scan_stats_launch(fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, int is_final, const fqh_carry *in,
                                 uint64_t *d_rec_start, uint64_t cap, uint32_t lmax, uint64_t *d_qual_hist,
                                 uint64_t *d_base_hist, uint64_t *d_scalars) {
    if (!ctx) return FQH_E_ARG;
    if (!d_qual_hist || !d_base_hist || !d_scalars || lmax == 0) return fail(ctx, FQH_E_ARG, "NULL histogram or lmax == 0");
    if (ctx->pending || ctx->stats_pending) return fail(ctx, FQH_E_ARG, "a launch is already pending");
    ctx->last_valid = false;  // never a reuse of an earlier scan: the offsets are wanted as well
    if (fqh_status hs = ensure_rows_hint(ctx, d_buf, len, in, lmax); hs != FQH_OK) return hs;
    if (fused_eligible(ctx, d_buf, len, is_final, in, lmax, 0, UINT64_MAX)) {
        fqh_status st = fused_launch(ctx, d_buf, len, is_final, in, d_rec_start, cap, lmax, d_qual_hist, d_base_hist, d_scalars);
        if (st != FQH_OK) return st;
        ctx->stats_pending = true;
        return FQH_OK;
    }
    count_down_backoffs(ctx);
    // two passes: the exact scan (offsets + full index), then the histogram kernel over that index
    const bool spec = ctx->spec_enabled;
    ctx->spec_enabled = false;
    fqh_status st = do_scan_launch(ctx, d_buf, len, is_final, in, d_rec_start, cap);
    if (st == FQH_OK) st = do_scan_finish(ctx, nullptr, nullptr);
    ctx->spec_enabled = spec;
    if (st != FQH_OK && st != FQH_E_CAPACITY) return st;
    // (FQH_E_CAPACITY: d_rec_start is too short; the histograms are counted all the same and fqh_scan_stats_finish reports it
    // with the exact summary, as the single-pass route does)
    const fqh_status cap_st = st;
    ctx->trust_index = true;   // (scan and count of one call)
    st = fqh_internal_stats_launch(ctx, d_buf, len, is_final, in, lmax, d_qual_hist, d_base_hist, d_scalars, 0, UINT64_MAX);
    ctx->trust_index = false;
    if (st == FQH_OK) ctx->stats_cap_st = cap_st;
    return st;
}
fqh_status fqh_scan_stats_finish(fqh_ctx *ctx, fqh_summary *out, fqh_carry *carry_out) {
    return fqh_stats_finish(ctx, out, carry_out);
}
fqh_status fqh_scan_stats(fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, int is_final, const fqh_carry *in,
                          uint64_t *d_rec_start, uint64_t cap, uint32_t lmax, uint64_t *d_qual_hist,
                          uint64_t *d_base_hist, uint64_t *d_scalars, fqh_summary *out, fqh_carry *carry_out) {
    fqh_status st = fqh_scan_stats_launch(ctx, d_buf, len, is_final, in, d_rec_start, cap, lmax, d_qual_hist, d_base_hist,
                                          d_scalars);
    if (st != FQH_OK) return st;
    return fqh_stats_finish(ctx, out, carry_out);
}

fqh_status fqh_stats(fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, int is_final, const fqh_carry *in,
                     uint32_t lmax, uint64_t *d_qual_hist, uint64_t *d_base_hist, uint64_t *d_scalars,
                     fqh_summary *out, fqh_carry *carry_out) {
    fqh_status st = fqh_stats_launch(ctx, d_buf, len, is_final, in, lmax, d_qual_hist, d_base_hist, d_scalars);
    if (st != FQH_OK) return st;
    return fqh_stats_finish(ctx, out, carry_out);
}

fqh_status fqh_len_hist(fqh_ctx *ctx, const uint64_t *d_base_hist, const uint64_t *d_scalars, uint32_t lmax,
                        uint64_t *d_len_hist) {
    if (!ctx || !d_base_hist || !d_scalars || !d_len_hist || lmax == 0) return FQH_E_ARG;
    if (ctx->pending || ctx->stats_pending) return fail(ctx, FQH_E_ARG, "a launch is pending");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    launch_len_hist(ctx->stream, (const unsigned long long *)d_base_hist, (const unsigned long long *)d_scalars, lmax,
                    (unsigned long long *)d_len_hist);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return FQH_OK;
}

fqh_status fqh_record_flags(fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, uint64_t base_offset,
                            const fqh_idx_record *d_index, uint64_t n, uint8_t *d_flags) {
    if (!ctx || (n && (!d_buf || !d_index || !d_flags))) return FQH_E_ARG;
    if (ctx->pending || ctx->stats_pending) return fail(ctx, FQH_E_ARG, "a launch is pending");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    launch_record_flags(ctx->stream, d_buf, len, base_offset, d_index, n, d_flags);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return FQH_OK;
}

fqh_status fqh_gather_records(fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, uint64_t base_offset,
                              const fqh_idx_record *d_index, uint64_t n, const uint8_t *d_flags, uint8_t mask,
                              uint8_t want, uint8_t *d_out, uint64_t out_cap, uint64_t *n_selected,
                              uint64_t *out_bytes) {
    if (!ctx || !n_selected || !out_bytes || (n && (!d_buf || !d_index || !d_flags))) return FQH_E_ARG;
    if (ctx->pending || ctx->stats_pending) return fail(ctx, FQH_E_ARG, "a launch is pending");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const uint64_t nb = gather_blocks(n);
    if (nb + 1 > ctx->gather_ws_blocks) {
        (void)hipFree(ctx->gather_ws);
        ctx->gather_ws = nullptr;
        ctx->gather_ws_blocks = 0;
        HIPCHK(ctx, hipMalloc((void **)&ctx->gather_ws, (2 * (nb + 1) + 2) * sizeof(unsigned long long)));
        ctx->gather_ws_blocks = nb + 1;
    }
    unsigned long long *bb = ctx->gather_ws, *br = bb + ctx->gather_ws_blocks, *tot = br + ctx->gather_ws_blocks;
    launch_gather(ctx->stream, d_buf, len, base_offset, d_index, n, d_flags, mask, want, bb, br, tot, d_out,
                  d_out ? out_cap : 0);
    HIPCHK(ctx, hipGetLastError());
    unsigned long long h[2] = {0, 0};
    HIPCHK(ctx, hipMemcpyAsync(h, tot, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *out_bytes = h[0];
    *n_selected = h[1];
    if (d_out && h[0] > out_cap) return fail(ctx, FQH_E_CAPACITY, "d_out capacity < bytes of the selected records");
    return FQH_OK;
}

}  // extern "C"
