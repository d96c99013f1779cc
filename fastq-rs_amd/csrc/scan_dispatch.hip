// scan_dispatch.hip — libfastq_hip.so, host side of the record scan (fqh_scan*, the byte-range shard protocol, fqh_index_records).
//
// Owns the per-context workspace in HBM (line-start lists, tile counts/prefixes), enqueues the
// kernels of scan_kernels.hip on one HIP stream, reads back one small struct
// per scan and turns it into the reference's observable result (record count, error kind, error
// record) — including the "Fastq record is too long" rule, which is a property of the reference's
// 68 KiB Buffer (src/buffer.rs:51-100, src/lib.rs:276-283) and is resolved here by replaying that
// buffer arithmetic over the record boundaries the GPU found.
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


// ---------------------------------------------------------------------------------------------
// with_list: the line lists (1 KiB per 16 KiB tile).  The exact path writes them; the fast path only needs them for tiles with
// more record starts than a tile's two lines hold (reads shorter than ~50 bp), so a context that only ever sees the fast
// path on ordinary reads never allocates them.
// Where the fast path's per-tile lines land in device memory decides how fast the byte scan runs: with one allocation
// k_index_fast takes 2.65-2.70 ms per 16 GiB, with another — same call, same size, same process — 2.85-2.95 ms, and the kind
// stays with the allocation for its lifetime (tools/exp_ctx_placement.py; the non-temporal line stores are what differs, the
// reads and every other kernel are the same).  Nothing visible from here tells the two kinds apart (virtual addresses do
// not), so the first big scan of a context tries: up to place_tries candidates, the index kernel timed on the first GiBs of
// the caller's own input with each, the fastest kept.  The measured difference is the only criterion; results do not depend
// on the choice.
static void place_fast_rs(fqh_ctx *ctx, size_t bytes) {
    const ScanArgs &a = ctx->args;
    const uint64_t full = a.len >> WT_SHIFT;
    if (ctx->place_tries < 2 || !a.buf || full < 65536 || !ctx->list_dummy || ((uintptr_t)a.buf & 15)) return;  // (small inputs: nothing to gain)
    const uint64_t st = full < 262144 ? full : 262144;  // whole tiles of the sample: up to 4 GiB
    DevOut *tmp_out = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipMalloc((void **)&tmp_out, sizeof(DevOut)) != hipSuccess) return;
    (void)hipMemsetAsync(tmp_out, 0, sizeof(DevOut), ctx->stream);
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        (void)hipFree(tmp_out);
        return;
    }
    auto timed = [&](auto launch) {  // three launches, the first warms up: the faster of the other two
        float m = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0, ctx->stream);
            launch();
            (void)hipEventRecord(e1, ctx->stream);
            (void)hipEventSynchronize(e1);
            float t = 0;
            if (hipEventElapsedTime(&t, e0, e1) == hipSuccess && rep > 0 && t < m) m = t;
        }
        return m;
    };
    // (the first handful of launches that store lines run 3-5 % slower than later ones, whatever memory they store to — seen
    // with candidates at different offsets of ONE allocation — so warm up before anything is compared)
    for (int w = 0; w < 6; ++w)
        launch_index(ctx->stream, a.buf, st << WT_SHIFT, ctx->list_dummy, 0u, ctx->tile_count, ctx->fast_rs, st, tmp_out, ctx->n_cu, true);
    // the yardstick: the same kernel over the same bytes without its line stores (no line buffer at all); with a line buffer of
    // the fast kind the stores cost 1-3 % on top of that, with one of the slow kind 7-11 %
    const float ceil_ms = timed([&] {
        launch_index(ctx->stream, a.buf, st << WT_SHIFT, ctx->list_dummy, 0u, ctx->tile_count, nullptr, st, tmp_out, ctx->n_cu, true);
    });
    constexpr int MAXC = 8;
    uint16_t *cand[MAXC] = {ctx->fast_rs, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    float ms[MAXC] = {0, 0, 0, 0, 0, 0, 0, 0};
    int n = 0, best = 0;
    const int tries = ctx->place_tries < MAXC ? ctx->place_tries : MAXC;
#ifdef FQH_TUNING  // experiment: candidates at different offsets of ONE allocation (is it the allocation or the address?)
    if (const char *e = getenv("FQH_WS_ARENA")) {
        const size_t step = (size_t)atoll(e) << 20;  // MiB between candidates
        uint8_t *arena = nullptr;
        if (hipMalloc((void **)&arena, bytes + 16 * step) == hipSuccess) {
            fprintf(stderr, "arena %p (input %p), without stores %.3f ms:", (void *)arena, (const void *)a.buf, ceil_ms);
            for (int k = 0; k < 16; ++k) {
                uint16_t *c = (uint16_t *)(arena + k * step);
                const float m = timed([&] { launch_index(ctx->stream, a.buf, st << WT_SHIFT, ctx->list_dummy, 0u, ctx->tile_count, c, st, tmp_out, ctx->n_cu, true); });
                fprintf(stderr, " +%zuM:%.3f", k * (step >> 20), m);
            }
            fprintf(stderr, "\n");
            (void)hipFree(arena);
        }
    }
#endif
    bool launch_err = hipGetLastError() != hipSuccess || ceil_ms > 1e29f;
    for (int k = 0; k < tries && !launch_err; ++k) {
        if (k > 0 && hipMalloc((void **)&cand[k], bytes) != hipSuccess) {
            cand[k] = nullptr;
            (void)hipGetLastError();
            break;
        }
        ms[k] = timed([&] {
            launch_index(ctx->stream, a.buf, st << WT_SHIFT, ctx->list_dummy, 0u, ctx->tile_count, cand[k], st, tmp_out, ctx->n_cu, true);
        });
        if (hipGetLastError() != hipSuccess || ms[k] > 1e29f) {  // a probe that did not run says nothing: keep what there was
            launch_err = true;
            if (k > 0) {
                (void)hipFree(cand[k]);
                cand[k] = nullptr;
            }
            break;
        }
        n = k + 1;
#ifdef FQH_TUNING  // validation of the criterion (tools/exp_ctx_placement.py): keep the SLOWEST candidate instead
        const bool worst = getenv("FQH_PLACE_PICK") && getenv("FQH_PLACE_PICK")[0] == 'w';
#else
        const bool worst = false;
#endif
        if (k > 0) {  // the loser of (best so far, this one) is freed at once: two line buffers at most are alive
            const bool better = worst ? ms[k] > ms[best] : ms[k] < ms[best];
            const int lose = better ? best : k;
            (void)hipFree(cand[lose]);
            cand[lose] = nullptr;
            if (better) best = k;
        }
        if (!worst && ms[best] <= 1.035f * ceil_ms) break;  // of the fast kind: no need to go on
    }
    for (int k = 0; k < n; ++k) ctx->place_ms[k] = ms[k];
    ctx->place_n = n;
    ctx->place_ms[8] = ms[best];
    ctx->place_ms[9] = ceil_ms;
    ctx->fast_rs = cand[best];
    if (getenv("FQH_DEBUG_WS")) {
        fprintf(stderr, "place_fast_rs: %d candidates over %.2f GiB (without stores %.3f ms):", n, (double)(st << WT_SHIFT) / (1 << 30), ceil_ms);
        for (int k = 0; k < n; ++k) fprintf(stderr, " %.3f%s", ms[k], k == best ? "*" : "");
        fprintf(stderr, " ms\n");
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(tmp_out);
    (void)hipGetLastError();
}

static fqh_status ensure_workspace(fqh_ctx *ctx, uint64_t n_tiles, bool with_list) {
    const size_t need_list = with_list ? (size_t)n_tiles * ctx->list_cap : 0;
    if (need_list > ctx->list_elems) {
        (void)hipFree(ctx->list);
        ctx->list = nullptr;
        ctx->list_elems = 0;
        HIPCHK(ctx, hipMalloc((void **)&ctx->list, std::max<size_t>(need_list, 1) * sizeof(uint16_t)));
        ctx->list_elems = need_list;
    }
    if (n_tiles > ctx->tiles_cap) {
        (void)hipFree(ctx->tile_count);
        (void)hipFree(ctx->tile_prefix);
        free_line_buffers(ctx);
        (void)hipFree(ctx->block_prefix);
        ctx->tile_count = ctx->tile_prefix = nullptr;
        ctx->fast_rs = nullptr;
        ctx->block_prefix = nullptr;
        ctx->tiles_cap = 0;
        const size_t nb = (n_tiles + SCAN_CHUNK - 1) / SCAN_CHUNK + 1;
        HIPCHK(ctx, hipMalloc((void **)&ctx->tile_count, n_tiles * sizeof(uint32_t)));
        HIPCHK(ctx, hipMalloc((void **)&ctx->tile_prefix, n_tiles * sizeof(uint32_t)));
        HIPCHK(ctx, hipMalloc((void **)&ctx->fast_rs, lines_bytes(n_tiles)));
        HIPCHK(ctx, hipMalloc((void **)&ctx->block_prefix, nb * sizeof(uint64_t)));
        ctx->tiles_cap = n_tiles;
        ctx->placed = false;  // (the first scan on the fast path tries placements of the new line buffer: place_fast_rs)
        if (getenv("FQH_DEBUG_WS")) fprintf(stderr, "workspace ctx %p: fast_rs %p tile_count %p tile_prefix %p block_prefix %p\n", (void *)ctx, (void *)ctx->fast_rs, (void *)ctx->tile_count, (void *)ctx->tile_prefix, (void *)ctx->block_prefix);
    }
    return FQH_OK;
}

// What turns the single pass's scratch into the caller's counts, all three conditional on DevOut::stats_commit (written by
// k_finalize_fast): the block partials and totals (k_stats_commit), the record in progress at the chunk start if the caller's
// buffer holds its beginning (+1), and the sequence line of the partial record behind the last complete one of a chunk that
// is not the file's last (-1): k_stats_edge.
void enqueue_fused_commit(fqh_ctx *ctx) {
    const ScanArgs &a = ctx->args;
    hipStream_t s = ctx->stream;
    unsigned long long *q = (unsigned long long *)ctx->f_qual, *b = (unsigned long long *)ctx->f_base, *sc = (unsigned long long *)ctx->f_scalars;
    launch_stats_commit(s, &ctx->d_out[0], ctx->f_args, scan_stats_blocks(a.n_tiles, ctx->n_cu), q, b, sc);
    launch_stats_declined(s, &ctx->d_out[0], ctx->f_args, q, b, sc);  // (batches with a byte outside the alphabets, lines beyond the rows; its set-up ran before the single pass was enqueued)
    const uint64_t back0 = a.back[a.nl_count & 3];
    if (back0 != 0 && ctx->f_lead >= back0) launch_stats_edge(s, &ctx->d_out[0], a.buf, a.len, back0, +1, ctx->f_lmax, q, b, sc);
    if (!a.is_final) launch_stats_edge(s, &ctx->d_out[0], a.buf, a.len, 0, -1, ctx->f_lmax, q, b, sc);
    ctx->f_commit_owed = false;
}

// (the buffer in use holds the tile index of the last scan — fqh_index_records, a rescan — and stays on the list until it is not)
static void adapt_free_rejects(fqh_ctx *ctx) {
    int keep = 0;
    for (int i = 0; i < ctx->n_rejects; ++i) {
        if (ctx->fr_rejects[i] == ctx->fast_rs) ctx->fr_rejects[keep++] = ctx->fr_rejects[i];
        else (void)hipFree(ctx->fr_rejects[i]);
    }
    ctx->n_rejects = keep;
}
// Which line buffer does this scan store to?  (ctx.h: LinesAdapt.)  Called for fresh fast-path scans of 2 GiB or more.
static void adapt_choose(fqh_ctx *ctx, bool fused) {
    const ScanArgs &a = ctx->args;
    ctx->adapt_entry = -1;
    // A new workspace (ensure_workspace has dropped every line buffer of the old one; fast_rs is the new allocation or the
    // placement search's pick): it is the primary.  NOT "fast_rs is neither of fr[]": after an alternate was given back, fast_rs
    // is that alternate — on the held-back list, fr[1] empty — and taking it for a new workspace lost fr[0] for good, one line
    // buffer of len / 64 bytes every four calls.
    if (!ctx->fr[0]) ctx->fr[0] = ctx->fast_rs;
    if (ctx->adapt_max <= 0 || a.len < (2ull << 30)) {
        ctx->fast_rs = ctx->fr[0];
        return;
    }
    int k = -1, lru = 0;
    for (int i = 0; i < 4; ++i) {
        if (ctx->adapt[i].buf == a.buf && ctx->adapt[i].len == a.len && ctx->adapt[i].fused == fused) k = i;
        if (ctx->adapt[i].stamp < ctx->adapt[lru].stamp) lru = i;
    }
    if (k < 0) {
        k = lru;
        ctx->adapt[k] = fqh_ctx::LinesAdapt{};
        ctx->adapt[k].buf = a.buf;
        ctx->adapt[k].len = a.len;
        ctx->adapt[k].fused = fused;
    }
    fqh_ctx::LinesAdapt &e = ctx->adapt[k];
    e.stamp = ++ctx->adapt_clock;
    int use = e.state == 3 ? e.choice : 0;
    if (e.state == 1) {
        if (!ctx->fr[1]) {
            const size_t bytes = lines_bytes(ctx->tiles_cap);
            if (hipMalloc((void **)&ctx->fr[1], bytes) != hipSuccess) {
                (void)hipGetLastError();
                ctx->fr[1] = nullptr;
                e.state = 3;
                e.choice = 0;
            }
        }
        if (ctx->fr[1]) use = 1;
    }
    if (use == 1 && !ctx->fr[1]) use = 0;
    ctx->fast_rs = ctx->fr[use];
    if (e.state < 3) {
        ctx->adapt_entry = k;
        ctx->adapt_used = use;
    }
}
// ... and what its index kernel took with it (HIP events of the launch: ctx->timing.index_ms)
static void adapt_record(fqh_ctx *ctx, bool clean) {
    const int k = ctx->adapt_entry;
    ctx->adapt_entry = -1;
    if (k < 0 || !clean || ctx->timing.index_ms <= 0) return;
    fqh_ctx::LinesAdapt &e = ctx->adapt[k];
    if (e.state == 0 && ctx->adapt_used == 0) {   // the first buffer is measured twice (the first big launch of a context runs cold)
        e.ms[0] = e.ms[0] > 0 ? std::min(e.ms[0], ctx->timing.index_ms) : ctx->timing.index_ms;
        if (++e.seen >= 2) e.state = 1;
        return;
    }
    if (e.state == 1 && ctx->adapt_used == 1) {   // ... and so is every alternate: one sample 2.5 % off is within the noise of a launch
        e.ms[1] = e.seen_alt ? std::min(e.ms[1], ctx->timing.index_ms) : ctx->timing.index_ms;
        if (++e.seen_alt < 2) return;
        e.seen_alt = 0;
    }
    if (e.state == 1 && ctx->adapt_used == 1) {
        const float a0 = e.ms[0], a1 = e.ms[1];
        if (getenv("FQH_DEBUG_WS")) fprintf(stderr, "adapt ctx %p input %p: first buffer %.3f ms, alternate %p %.3f ms (try %d)\n", (void *)ctx, (const void *)e.buf, a0, (void *)ctx->fr[1], a1, e.tries);
        if (a1 < 0.975f * a0 || a0 < 0.975f * a1) {   // the two buffers are of different kinds for this input: keep both, take the faster
            e.choice = a1 < a0 ? 1 : 0;
            e.state = 3;
            adapt_free_rejects(ctx);
        } else {                                      // alike: this alternate tells nothing.  Give it back unless another input has chosen it
            bool wanted = false;
            for (const auto &o : ctx->adapt) wanted = wanted || (&o != &e && o.buf && o.state == 3 && o.choice == 1);
            ++e.tries;
            if (!wanted && e.tries < ctx->adapt_max && ctx->n_rejects < 8) {
                // held, not freed, until this input is settled: a free followed by an allocation of the same size hands the
                // same memory out again (and the tile index of the scan just finished lives in it)
                ctx->fr_rejects[ctx->n_rejects++] = ctx->fr[1];
                ctx->fr[1] = nullptr;
                e.state = 1;           // the next call allocates another one
            } else {
                e.choice = 0;
                e.state = 3;
                adapt_free_rejects(ctx);
            }
        }
    }
}

// fast == true: the speculative path (k_index_fast + k_emit_fast + k_finalize_fast)
static fqh_status enqueue_scan(fqh_ctx *ctx, bool reuse_index, bool fast) {
    ScanArgs &a = ctx->args;
    hipStream_t s = ctx->stream;
    const bool with_list = !fast || ctx->fast_needs_list;
    fqh_status st = ensure_workspace(ctx, a.n_tiles, with_list);
    if (st != FQH_OK) return st;
    if (fast && !ctx->placed) {
        place_fast_rs(ctx, lines_bytes(ctx->tiles_cap));
        ctx->placed = true;
    }
    if (fast && !reuse_index) adapt_choose(ctx, ctx->fused);
    a.list = with_list ? ctx->list : nullptr;
    a.list_cap = ctx->list_cap;
    a.tile_count = ctx->tile_count;
    a.fast_rs = lines_in_use(ctx);
    a.tile_prefix = ctx->tile_prefix;
    a.block_prefix = ctx->block_prefix;
    ctx->used_spec = fast;
    a.mirror = ctx->h_out;  // the finalize kernel publishes its results there and resets the accumulators
    if (!ctx->dout_clean) HIPCHK(ctx, hipMemcpyAsync(&ctx->d_out[0], ctx->h_init, sizeof(DevOut), hipMemcpyHostToDevice, s));
    ctx->dout_clean = false;
    HIPCHK(ctx, hipEventRecord(ctx->ev[0], s));
    FusedArgs fz = {};
    const bool fused = fast && ctx->fused && !reuse_index;
    if (fused) {
        // the single-pass kernel: byte scan of the fast path + histograms (fused_kernels.hip).  Its partial
        // histograms and totals go to scratch; k_stats_commit (below) adds them to the caller's arrays if the
        // finalize kernel keeps the fast path's result.
        const size_t need = scan_stats_scratch_bytes(ctx->n_cu);
        if (need > ctx->stats_scratch_bytes) {
            (void)hipFree(ctx->stats_scratch);
            ctx->stats_scratch = nullptr;
            ctx->stats_scratch_bytes = 0;
            HIPCHK(ctx, hipMalloc((void **)&ctx->stats_scratch, need));
            ctx->stats_scratch_bytes = need;
        }
        if (!ctx->side) HIPCHK(ctx, hipMalloc((void **)&ctx->side, 2 * FQH_NSCALARS * sizeof(unsigned long long)));
        {   // where the kernel puts what it will not count itself: a slot per 512 KiB of input + 1024 (1.5 KiB per slot: 0.3 % of the input)
            const uint64_t cap = a.len / (512u << 10) + 1024;
            const size_t bb = (size_t)cap * (1 + scan_stats_nsl(ctx->f_rows)) * 64 * sizeof(uint32_t);
            if (cap > ctx->decl_cap || bb > ctx->decl_b_bytes) {
                (void)hipFree(ctx->decl_b);
                (void)hipFree(ctx->decl_l);
                ctx->decl_b = nullptr;
                ctx->decl_l = nullptr;
                ctx->decl_cap = 0;
                ctx->decl_b_bytes = 0;
                HIPCHK(ctx, hipMalloc((void **)&ctx->decl_b, bb));
                HIPCHK(ctx, hipMalloc((void **)&ctx->decl_l, (size_t)cap * 2 * sizeof(uint64_t)));
                ctx->decl_cap = (uint32_t)cap;
                ctx->decl_b_bytes = bb;
            }
        }
        HIPCHK(ctx, hipMemsetAsync(ctx->side, 0, 2 * FQH_NSCALARS * sizeof(unsigned long long), s));
        HIPCHK(ctx, hipEventRecord(ctx->ev[0], s));  // (after the memset: index_ms is the kernel alone)
        fz.buf = a.buf;
        fz.len = a.len;
        fz.n_tiles = a.n_tiles;
        fz.list = const_cast<uint16_t *>(a.list);
        fz.list_cap = ctx->list_cap;
        fz.fast_rs = lines_in_use(ctx);
        fz.out = &ctx->d_out[0];
        fz.lmax = ctx->f_lmax;
        fz.rows = ctx->f_rows;
        fz.scratch = ctx->stats_scratch;
        fz.scalars = ctx->side;
        fz.skip_head = a.back[a.nl_count & 3] != 0 ? 1u : 0u;  // the chunk begins inside a record: that one is k_stats_edge's
        fz.decl_b = ctx->decl_b;
        fz.decl_l = ctx->decl_l;
        fz.decl_cap = ctx->decl_cap;
        ctx->f_args = fz;
        HIPCHK(ctx, prepare_stats_declined(fz.rows));  // (may fail; nothing of this call is enqueued yet that could commit without it)
        if (a.n_tiles) HIPCHK(ctx, launch_scan_stats(s, fz, ctx->n_cu));
#ifdef FQH_TUNING
        if (getenv("FQH_FZ_WHY")) {
            unsigned long long w = 0;
            (void)hipMemcpyAsync(&w, ctx->side + 15, sizeof w, hipMemcpyDeviceToHost, s);
            (void)hipStreamSynchronize(s);
            fprintf(stderr, "FZ_WHY %llu\n", w);
        }
#endif
#ifdef FQH_FZ_TIMING  // tuning builds only: cycles a wave spends per phase of a group (tools/exp_fztime.sh)
        {
            unsigned long long t[8];
            (void)hipMemcpyAsync(t, ctx->side + FQH_NSCALARS, sizeof t, hipMemcpyDeviceToHost, s);
            (void)hipStreamSynchronize(s);
            unsigned long long tt = 0;
            for (int i = 0; i < 8; ++i) tt += t[i];
            fprintf(stderr, "FZ_TIMING %% of a group's cycles:");
            for (int i = 0; i < 8; ++i) fprintf(stderr, " [%d] %.1f", i, tt ? 100.0 * (double)t[i] / (double)tt : 0.0);
            fprintf(stderr, "  (cycles per group and wave: %.0f)\n", (double)tt / ((double)fz.len / 4096.0));
        }
#endif
        ctx->index_full = false;
    } else if (!reuse_index) {
        // (the fast path without a line-list workspace: what a tile with more than 116 record starts would put there goes
        // to a 1 KiB dummy — list_cap 0 — and k_emit_fast reports such a tile, need_list)
        launch_index(s, a.buf, a.len, a.list ? const_cast<uint16_t *>(a.list) : ctx->list_dummy, a.list ? ctx->list_cap : 0u,
                     ctx->tile_count, const_cast<uint16_t *>(a.fast_rs), a.n_tiles, &ctx->d_out[0], ctx->n_cu, fast);
        ctx->index_full = !fast;
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev[1], s));
    if (!reuse_index)
        launch_prefix(s, ctx->tile_count, fast ? a.fast_rs : nullptr, ctx->tile_prefix, ctx->block_prefix, a.n_tiles,
                      a.n_blocks);
    HIPCHK(ctx, hipEventRecord(ctx->ev[2], s));
    if (fast) {
        if (!ctx->skip_emit) launch_emit_fast(s, a, &ctx->d_out[0], ctx->n_cu);
        launch_finalize_fast(s, a, &ctx->d_out[0]);  // a prescan still needs the newline count and the carry
        if (fused && a.n_tiles) {
            ctx->f_commit_owed = ctx->f_defer_commit;
            if (!ctx->f_defer_commit) enqueue_fused_commit(ctx);
        }
    } else {
        if (!ctx->skip_emit) launch_emit(s, a, &ctx->d_out[0], ctx->n_cu);
        launch_finalize(s, a, &ctx->d_out[0]);
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev[3], s));
    HIPCHK(ctx, hipGetLastError());
    return FQH_OK;
}


// The cached tile index describes the BYTES of the last scanned buffer: every writer of this library that touches
// them drops it.  (Writes the library cannot see — the caller's own kernels, a reused allocator block — need
// fqh_invalidate; see include/fastq_hip.h.)
void drop_index_if_overlaps(fqh_ctx *ctx, const void *d_dst, uint64_t bytes) {
    const uint8_t *lo = (const uint8_t *)d_dst, *hi = lo + bytes;
    if (ctx->last_valid && lo < ctx->args.buf + ctx->args.len && hi > ctx->args.buf) ctx->last_valid = false;
}
static bool carry_is_zero(const fqh_carry &c) {
    return c.base_offset == 0 && c.nl_count == 0 && c.back[0] == 0 && c.back[1] == 0 && c.back[2] == 0 && c.back[3] == 0;
}

static fqh_status resolve(fqh_ctx *ctx, fqh_summary *out, fqh_carry *carry_out) {
    const ScanArgs &a = ctx->args;
    const DevOut &d = *ctx->h_out;
    fqh_summary s = {};
    const uint64_t r0 = a.nl_count >> 2;
    s.n_records = d.n_records;
    s.bytes_consumed = d.end_off > 0 ? (uint64_t)d.end_off : 0;
    s.parse_status = FQH_OK;
    s.n_newlines = d.n_newlines;
    s.tail_len = d.tail_len;
    s.max_record_len = d.max_len >= (1ull << 60) ? UINT64_MAX : d.max_len;
    s.n_line_starts = d.total_entries + d.lastnl;
    if (d.final_key != NOKEY) {
        static const int32_t stage_status[4] = {FQH_E_HEADER, FQH_E_SEP, FQH_E_LEN_MISMATCH, FQH_E_TRUNCATED};
        s.parse_status = stage_status[d.final_key & 3];
        s.err_record = d.final_key >> 2;
        s.err_offset = a.base_offset + (uint64_t)d.err_start;
    }
    // "Fastq record is too long" (src/lib.rs:278-283) depends on a record's file offset mod 16 and on nothing else (the closed form
    // of the reference's Buffer arithmetic, csrc/replay.h: fqh::TooLong), so a CHUNK with a carry is judged like a whole file —
    // its boundaries are file offsets.  Not for launches whose carry is made up (prescans, fqh_shard_align) and not for the
    // ring's slots (the ring applies the same rule to the boundaries it downloads anyway, and holds its commits back for it).
    if (ctx->bufsize && ctx->launch_long_rule && !a.prescan) {
        const uint64_t B = ctx->bufsize;
        bool cand = d.first_long != NOKEY && d.first_long < r0 + s.n_records;
        uint64_t need = TooLong::NO_BAD;
        if (d.final_key != NOKEY) {
            const uint32_t stage = (uint32_t)d.final_key & 3u;
            if (stage == 3) {
                need = 0;
                if (a.len - (uint64_t)d.err_start + 15 >= B) cand = true;
            } else {
                need = d.err_need;
                if (need + 15 > B) cand = true;
            }
        } else if (d.tail_len + 15 >= B) {
            cand = true;   // the record in progress at the end of a chunk that is not the file's last
        }
        if (cand) {
            // record boundaries on the host
            const uint64_t n = s.n_records;
            std::vector<uint64_t> rs(n + 1);
            const uint64_t *src = a.rec_start;
            if (!src || a.cap < n + 1) {
                if (ctx->tmp_rec_cap < n + 1) {
                    (void)hipFree(ctx->tmp_rec);
                    ctx->tmp_rec = nullptr;
                    ctx->tmp_rec_cap = 0;
                    HIPCHK(ctx, hipMalloc((void **)&ctx->tmp_rec, (n + 1) * sizeof(uint64_t)));
                    ctx->tmp_rec_cap = n + 1;
                }
                fqh_status fst = ensure_full_index(ctx);
                if (fst != FQH_OK) return fst;
                ScanArgs b = ctx->args;
                b.mirror = nullptr;  // a side launch on d_out[1]: the main results in h_out stay
                b.rec_start = ctx->tmp_rec;
                b.cap = n + 1;
                b.idx = nullptr;
                HIPCHK(ctx, hipMemcpyAsync(&ctx->d_out[1], ctx->h_init, sizeof(DevOut), hipMemcpyHostToDevice, ctx->stream));
                launch_emit(ctx->stream, b, &ctx->d_out[1], ctx->n_cu);
                launch_finalize(ctx->stream, b, &ctx->d_out[1]);
                src = ctx->tmp_rec;
            }
            HIPCHK(ctx, hipMemcpyAsync(rs.data(), src, (n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            rs[n] = (uint64_t)((long long)a.base_offset + d.end_off);  // authoritative end of the last good record (a file offset, like the others)
            uint64_t which = 0;
            if (TooLong::first(B, rs.data(), n, a.base_offset + a.len - rs[n], need, &which)) {
                s.parse_status = FQH_E_TOO_LONG;
                s.n_records = which;
                s.err_record = r0 + which;
                s.err_offset = rs[which];
                s.bytes_consumed = which && rs[which] > a.base_offset ? rs[which] - a.base_offset : 0;
            }
        }
    }
    fqh_carry c = {};
    c.base_offset = a.base_offset + a.len;
    c.nl_count = a.nl_count + d.n_newlines;
    for (int i = 0; i < 4; ++i) {
        long long v = (long long)a.len - d.recent[i];
        uint64_t lim = a.base_offset + a.len;
        c.back[i] = v < 0 ? 0 : ((uint64_t)v > lim ? lim : (uint64_t)v);
    }
    ctx->last_summary = s;
    ctx->last_carry_out = c;
    // (what every scan says about the reads' length: no sequence / quality line of a delivered record is longer than half the
    // longest record — the single pass's rows may come DOWN to that, whichever route found it; they go up in update_rows_hint)
    // ... of the input the belief is about (a plain scan of ANOTHER buffer in between says nothing about it)
    const bool hinted = ctx->hint_valid && a.buf == ctx->hint_buf && a.len == ctx->hint_len && a.base_offset == ctx->hint_base;
    if (hinted && s.n_records && s.max_record_len / 2 && s.max_record_len / 2 < ctx->rows_hint) ctx->rows_hint = (uint32_t)(s.max_record_len / 2);
    if (hinted && s.n_records && s.max_record_len / 2 <= 511) ctx->lines_long = false;
    if (out) *out = s;
    if (carry_out) *carry_out = c;
    // (whatever the parse status: the emit kernels clamp their writes to cap, so a caller that walks
    // d_rec_start[0 .. n_records] of a chunk with an error AND more records than cap would read past it;
    // summary and carry are exact all the same)
    if (a.rec_start && s.n_records + 1 > a.cap) return fail(ctx, FQH_E_CAPACITY, "d_rec_start capacity < n_records + 1");
    return FQH_OK;
}

fqh_status do_scan_launch(fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, int is_final, const fqh_carry *in,
                          uint64_t *d_rec_start, uint64_t cap, bool reuse_index) {
    if (!ctx) return FQH_E_ARG;
    if (ctx->pending || ctx->stats_pending) return fail(ctx, FQH_E_ARG, "a launch is already pending");
    if (len && !d_buf) return fail(ctx, FQH_E_ARG, "d_buf is NULL");
    if (len && ((uintptr_t)d_buf & 15)) return fail(ctx, FQH_E_ARG, "d_buf must be 16-byte aligned");
    if (d_rec_start && cap == 0) return fail(ctx, FQH_E_ARG, "cap is 0");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->last_valid = false;
    fqh_carry c = {};
    if (in) c = *in;
    for (int i = 0; i < 4; ++i)
        if (c.back[i] > c.base_offset) return fail(ctx, FQH_E_ARG, "carry.back exceeds base_offset");
    ctx->carry_in = c;
    ctx->whole_file = is_final && carry_is_zero(c);
    ctx->launch_long_rule = !ctx->no_long_rule;
    ScanArgs &a = ctx->args;
    a = ScanArgs{};
    a.buf = d_buf;
    a.len = len;
    a.base_offset = c.base_offset;
    a.nl_count = c.nl_count;
    for (int i = 0; i < 4; ++i) a.back[i] = c.back[i];
    a.is_final = is_final ? 1 : 0;
    a.v_start = (c.back[0] == 0 && len > 0) ? 1u : 0u;
    a.bufsize = ctx->bufsize;
    a.max_walk = ctx->bufsize ? (uint32_t)(ctx->bufsize / WT_BYTES + 3) : 0xFFFFFFFFu;
    a.head_unchecked = ctx->head_unchecked ? 1u : 0u;
    a.prescan = ctx->skip_emit ? 1u : 0u;
    a.n_tiles = (len + WT_BYTES - 1) / WT_BYTES;
    a.n_blocks = (a.n_tiles + SCAN_CHUNK - 1) / SCAN_CHUNK;
    a.rec_start = d_rec_start;
    a.cap = d_rec_start ? cap : 0;
    a.idx = ctx->scan_idx;       // (a statistics call over kilobase reads asks the scan's own emit step for the record index)
    a.idx_cap = ctx->scan_idx ? ctx->scan_idx_cap : 0;
    ctx->idx_emitted = ctx->scan_idx != nullptr;
    // fast path: when the caller does not need full line lists and no earlier input needed the exact
    // path; a rescan on a retained index uses whichever kind of index is there
    bool fast = reuse_index ? !ctx->index_full : (ctx->spec_enabled && !ctx->exact_holds && ctx->list_cap >= LIST_CAP_DEFAULT);
    if (fast && !reuse_index && ctx->spec_skip) {  // backing off after a failed attempt
        --ctx->spec_skip;
        fast = false;
    }
    fqh_status st = enqueue_scan(ctx, reuse_index, fast);
    if (st != FQH_OK) return st;
    ctx->pending = true;
    return FQH_OK;
}

fqh_status do_scan_finish(fqh_ctx *ctx, fqh_summary *out, fqh_carry *carry_out) {
    if (!ctx) return FQH_E_ARG;
    if (!ctx->pending) return fail(ctx, FQH_E_ARG, "no scan pending");
    ctx->pending = false;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (ctx->spin_wait_us > 0) {
        // FQH_OPT_SPIN_WAIT (off by default: a host core spinning inside a library call is the caller's decision): poll the
        // stream for up to that many microseconds before sleeping on it — hipStreamSynchronize wakes up ~15 us after the
        // last kernel, 0.5 % of a 16 GiB step.
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t q;
        while ((q = hipStreamQuery(ctx->stream)) == hipErrorNotReady &&
               std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(ctx->spin_wait_us)) {}
        (void)hipGetLastError();
        if (q != hipSuccess) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    } else {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    ctx->dout_clean = true;  // the finalize kernel has run
    const int adapt_k = ctx->adapt_entry;   // (a rerun below measures something else: only a launch that stood is recorded)
    int reruns = 0;
    if (ctx->dev_carry) {  // the carry was folded on the device (fqh_shard_rescan_launch): the host learns it here
        ctx->dev_carry = false;
        const DevCarry hc = *ctx->h_carry;
        ScanArgs &a = ctx->args;
        a.dcarry = nullptr;
        a.base_offset = hc.base_offset;
        a.nl_count = hc.nl_count;
        for (int i = 0; i < 4; ++i) a.back[i] = hc.back[i];
        a.v_start = (hc.back[0] == 0 && a.len > 0) ? 1u : 0u;
        ctx->carry_in.base_offset = hc.base_offset;
        ctx->carry_in.nl_count = hc.nl_count;
        for (int i = 0; i < 4; ++i) ctx->carry_in.back[i] = hc.back[i];
        if (hc.any_fail) {
            ctx->last_valid = false;
            return fail(ctx, FQH_E_AGAIN, "a shard's byte scan left the fast path: run fqh_shard_prescan / fqh_carry_combine / fqh_rescan_launch");
        }
    }
    if (ctx->used_spec && ctx->h_out->spec_fail && ctx->h_out->need_list && !ctx->fast_needs_list) {
        // not a doubt about the input: a tile holds more record starts than its two lines take (reads shorter than ~50 bp)
        // and this context has no line-list workspace yet.  Allocate it and run the fast path again; it stays.
        ctx->fast_needs_list = true;
        ++reruns;
        fqh_status st = enqueue_scan(ctx, false, true);
        if (st != FQH_OK) return st;
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        ctx->dout_clean = true;
    }
#ifdef FQH_TUNING  // knock-out timings (tools/exp_fz2.sh): the result is wrong by design, keep the fast path's timing
    if (getenv("FQH_FZ_DBG") && atoi(getenv("FQH_FZ_DBG")) != 0) ctx->h_out->spec_fail = 0;
#endif
    if (ctx->used_spec && ctx->h_out->spec_fail) {
        // the fast path could not prove the input valid (a real error, lines longer than a few KiB,
        // or a degenerate layout): run the exact path, and keep later scans of this context on it for a
        // while (1, 2, 4 .. 64 scans: one bad file should not cost the fast path for good, a stream of
        // inputs the fast path cannot handle should not pay for an attempt every time)
        ctx->spec_backoff = ctx->spec_backoff ? (ctx->spec_backoff < 64 ? ctx->spec_backoff * 2 : 64) : 1;
        ctx->spec_skip = ctx->spec_backoff;
        ++reruns;
        fqh_status st = enqueue_scan(ctx, false, false);
        if (st != FQH_OK) return st;
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        ctx->dout_clean = true;
    }
    else if (ctx->used_spec) ctx->spec_backoff = 0;
    // (ctx->used_spec now tells whether the result in h_out came from the fast path)
    if (ctx->h_out->overflow) {
        // a tile has more line starts than list_cap (lines shorter than 32 bytes on average): rerun
        // with longer lists; the setting sticks to the context, so steady state stays single-pass.
        while (ctx->h_out->overflow) {
            if (ctx->list_cap >= WT_BYTES) return fail(ctx, FQH_E_DEVICE, "line list overflow with full-size lists");
            ctx->list_cap = ctx->list_cap < 2048 ? 2048 : WT_BYTES;
            ++reruns;
            fqh_status st = enqueue_scan(ctx, false, false);
            if (st != FQH_OK) return st;
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            ctx->dout_clean = true;
        }
    }
    float ms = 0;
    ctx->timing = fqh_timing{};
    if (hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]) == hipSuccess) ctx->timing.index_ms = ms;
    if (hipEventElapsedTime(&ms, ctx->ev[1], ctx->ev[2]) == hipSuccess) ctx->timing.prefix_ms = ms;
    if (hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]) == hipSuccess) ctx->timing.emit_ms = ms;
    if (hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[3]) == hipSuccess) ctx->timing.total_ms = ms;
    fqh_status st = resolve(ctx, out, carry_out);
    ctx->last_valid = (st == FQH_OK || st == FQH_E_CAPACITY);
    {
        ctx->adapt_entry = adapt_k;
        adapt_record(ctx, adapt_k >= 0 && ctx->used_spec && reruns == 0 && !ctx->h_out->stats_declined);
    }
    ctx->fused = false;  // (the launch is over; a deferred commit, f_commit_owed, stays owed if the fast path stood)
    ctx->f_defer_commit = false;
    if (!ctx->used_spec) ctx->f_commit_owed = false;
    return st;
}

// May a statistics call count over the tile index of the last finished scan instead of scanning again?  Only when the
// caller has said that the bytes are what they were (FQH_OPT_REUSE_INDEX, or the library's own scan-then-count sequences:
// the ring): matching pointer, length and carry do not prove it — the caller's own kernels may have rewritten the buffer.
bool same_scan(const fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, int is_final, const fqh_carry *in) {
    if (!ctx->last_valid || !(ctx->reuse_index || ctx->trust_index)) return false;
    fqh_carry c = {};
    if (in) c = *in;
    return ctx->args.buf == d_buf && ctx->args.len == len && ctx->args.is_final == (is_final ? 1 : 0) &&
           memcmp(&c, &ctx->carry_in, sizeof c) == 0;
}

// index-only emit of the last scan into `dst` (device), local records [0, n)
// (re)build complete line lists for the buffer of the last scan (after a fast-path scan)
fqh_status ensure_full_index(fqh_ctx *ctx) {
    if (ctx->index_full) return FQH_OK;
    const ScanArgs &a = ctx->args;
    for (;;) {
        fqh_status st = ensure_workspace(ctx, a.n_tiles, true);
        if (st != FQH_OK) return st;
        HIPCHK(ctx, hipMemcpyAsync(&ctx->d_out[1], ctx->h_init, sizeof(DevOut), hipMemcpyHostToDevice, ctx->stream));
        launch_index(ctx->stream, a.buf, a.len, ctx->list, ctx->list_cap, ctx->tile_count, ctx->fast_rs,
                     a.n_tiles, &ctx->d_out[1], ctx->n_cu, false);
        launch_prefix(ctx->stream, ctx->tile_count, nullptr, ctx->tile_prefix, ctx->block_prefix, a.n_tiles, a.n_blocks);
        DevOut tmp;
        HIPCHK(ctx, hipMemcpyAsync(&tmp, &ctx->d_out[1], sizeof(DevOut), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (!tmp.overflow) break;
        if (ctx->list_cap >= WT_BYTES) return fail(ctx, FQH_E_DEVICE, "line list overflow with full-size lists");
        ctx->list_cap = ctx->list_cap < 2048 ? 2048 : WT_BYTES;
    }
    ctx->index_full = true;
    ctx->args.list = ctx->list;
    ctx->args.list_cap = ctx->list_cap;
    ctx->args.tile_count = ctx->tile_count;
    ctx->args.tile_prefix = ctx->tile_prefix;
    ctx->args.block_prefix = ctx->block_prefix;
    return FQH_OK;
}

fqh_status emit_index(fqh_ctx *ctx, fqh_idx_record *dst, uint64_t cap) {
    fqh_status fst = ensure_full_index(ctx);
    if (fst != FQH_OK) return fst;
    ScanArgs b = ctx->args;
    b.mirror = nullptr;  // a side launch on d_out[1]: the main results in h_out stay
    b.rec_start = nullptr;
    b.cap = 0;
    b.idx = dst;
    b.idx_cap = cap;
    HIPCHK(ctx, hipMemcpyAsync(&ctx->d_out[1], ctx->h_init, sizeof(DevOut), hipMemcpyHostToDevice, ctx->stream));
    launch_emit(ctx->stream, b, &ctx->d_out[1], ctx->n_cu);
    launch_finalize(ctx->stream, b, &ctx->d_out[1]);
    HIPCHK(ctx, hipGetLastError());
    return FQH_OK;
}

fqh_status fqh_internal_scan_launch(fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, int is_final,
                                    const fqh_carry *in, uint64_t *d_rec_start, uint64_t cap, bool reuse_index) {
    return do_scan_launch(ctx, d_buf, len, is_final, in, d_rec_start, cap, reuse_index);
}
fqh_status fqh_internal_scan_finish(fqh_ctx *ctx, fqh_summary *out, fqh_carry *carry_out) {
    return do_scan_finish(ctx, out, carry_out);
}
fqh_status fqh_internal_emit_index(fqh_ctx *ctx, fqh_idx_record *dst, uint64_t cap) { return emit_index(ctx, dst, cap); }
uint64_t fqh_internal_last_need(const fqh_ctx *ctx) {
    const DevOut &d = *ctx->h_out;
    if (d.final_key == NOKEY) return BufferReplay::NO_BAD;
    return ((uint32_t)d.final_key & 3u) == 3 ? 0 : d.err_need;
}

extern "C" {

fqh_status fqh_scan_launch(fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, int is_final,
                           const fqh_carry *in, uint64_t *d_rec_start, uint64_t cap) {
    return do_scan_launch(ctx, d_buf, len, is_final, in, d_rec_start, cap);
}
fqh_status fqh_scan_finish(fqh_ctx *ctx, fqh_summary *out, fqh_carry *carry_out) {
    return do_scan_finish(ctx, out, carry_out);
}
fqh_status fqh_scan(fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, int is_final, const fqh_carry *in,
                    uint64_t *d_rec_start, uint64_t cap, fqh_summary *out, fqh_carry *carry_out) {
    fqh_status st = do_scan_launch(ctx, d_buf, len, is_final, in, d_rec_start, cap);
    if (st != FQH_OK) return st;
    return do_scan_finish(ctx, out, carry_out);
}

fqh_status fqh_shard_prescan(fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, uint64_t *n_newlines,
                             uint64_t *n_line_starts, uint64_t back0[4]) {
    if (!ctx || !n_newlines || !n_line_starts || !back0) return FQH_E_ARG;
    ctx->skip_emit = true;
    fqh_status st = do_scan_launch(ctx, d_buf, len, 0, nullptr, nullptr, 0);
    ctx->skip_emit = false;
    if (st != FQH_OK) return st;
    fqh_summary s;
    fqh_carry c;
    ctx->skip_emit = true;  // an overflow rerun inside finish must skip the emit as well
    st = do_scan_finish(ctx, &s, &c);
    ctx->skip_emit = false;
    if (st != FQH_OK) return st;
    *n_newlines = s.n_newlines;
    *n_line_starts = s.n_line_starts;
    for (int i = 0; i < 4; ++i) back0[i] = c.back[i];
    return FQH_OK;
}

fqh_status fqh_shard_prescan_launch(fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, uint64_t *d_words) {
    if (!ctx || !d_words) return FQH_E_ARG;
    ctx->skip_emit = true;
    fqh_status st = do_scan_launch(ctx, d_buf, len, 0, nullptr, nullptr, 0);
    ctx->skip_emit = false;
    if (st != FQH_OK) return st;
    launch_shard_words(ctx->stream, &ctx->d_out[0], ctx->h_out, len, d_words);
    HIPCHK(ctx, hipGetLastError());
    return FQH_OK;  // (the launch stays pending: fqh_shard_rescan_launch continues it, fqh_scan_finish ends it)
}

fqh_status fqh_shard_rescan_launch(fqh_ctx *ctx, int is_final, const uint64_t *d_all_words, int n_ranks, int rank,
                                   uint64_t *d_rec_start, uint64_t cap, uint64_t *d_counts) {
    if (!ctx || !d_all_words || n_ranks < 1 || rank < 0 || rank >= n_ranks) return FQH_E_ARG;
    if (!ctx->pending || ctx->dev_carry || !ctx->args.prescan)
        return fail(ctx, FQH_E_ARG, "fqh_shard_rescan_launch continues a fqh_shard_prescan_launch");
    if (d_rec_start && cap == 0) return fail(ctx, FQH_E_ARG, "cap is 0");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!ctx->d_carry) HIPCHK(ctx, hipMalloc((void **)&ctx->d_carry, sizeof(DevCarry)));
    if (!ctx->h_carry) HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_carry, sizeof(DevCarry), hipHostMallocDefault));
    hipStream_t s = ctx->stream;
    launch_carry_fold(s, d_all_words, n_ranks, rank, ctx->d_carry, ctx->h_carry, &ctx->d_out[0]);
    ScanArgs &a = ctx->args;  // the prescan's: buffer and tile index
    a.is_final = is_final ? 1 : 0;
    a.rec_start = d_rec_start;
    a.cap = d_rec_start ? cap : 0;
    a.dcarry = ctx->d_carry;
    a.prescan = 0;
    ctx->whole_file = false;
    ctx->launch_long_rule = true;  // (the folded carry holds the shard's true file offset: "too long" is judged on it, as for any chunk)
    ctx->dev_carry = true;
    ctx->dout_clean = false;
    HIPCHK(ctx, hipEventRecord(ctx->ev[2], s));
    if (ctx->used_spec) {
        launch_emit_fast(s, a, &ctx->d_out[0], ctx->n_cu);
        launch_finalize_fast(s, a, &ctx->d_out[0]);
    } else {
        launch_emit(s, a, &ctx->d_out[0], ctx->n_cu);
        launch_finalize(s, a, &ctx->d_out[0]);
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev[3], s));
    if (d_counts) launch_shard_counts(s, &ctx->d_out[0], ctx->h_out, ctx->d_carry, d_counts);
    HIPCHK(ctx, hipGetLastError());
    return FQH_OK;
}

fqh_status fqh_shard_align(fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, int prev_is_newline, uint32_t *phase,
                           uint64_t *first_record_offset) {
    if (!ctx || !phase || !first_record_offset || !len) return FQH_E_ARG;
    *phase = 0xFFFFFFFFu;
    *first_record_offset = 0;
    // index the window once (phase-free), then run only the emit / validate step under each of the four phases
    uint64_t nn = 0, ns = 0, back0[4];
    fqh_status st = fqh_shard_prescan(ctx, d_buf, len, &nn, &ns, back0);
    if (st != FQH_OK) return st;
    int found = -1, n_ok = 0;
    uint64_t off = 0;
    uint64_t recs[4] = {0, 0, 0, 0}, offs[4] = {0, 0, 0, 0};
    bool clean[4] = {false, false, false, false};
    const uint32_t keep_skip = ctx->spec_skip, keep_backoff = ctx->spec_backoff;  // (three of the four probes fail by design)
    for (uint32_t phi = 0; phi < 4; ++phi) {
        fqh_carry c = {};
        c.base_offset = 1ull << 40;  // (anywhere: only differences to it are used)
        c.nl_count = phi;
        // distances to the line starts before the window: unknown.  Anything consistent will do, the record in
        // progress is not validated (head_unchecked); back[0] == 0 says "a line starts at offset 0"
        c.back[0] = prev_is_newline ? 0 : 1;
        for (int i = 1; i < 4; ++i) c.back[i] = c.back[i - 1] + 1;
        ctx->head_unchecked = true;
        ctx->no_long_rule = true;   // (a made-up base offset: nothing to judge "too long" on)
        st = do_scan_launch(ctx, d_buf, len, 0, &c, (uint64_t *)ctx->d_misc, 2, true);
        ctx->no_long_rule = false;
        fqh_summary s = {};
        if (st == FQH_OK) st = do_scan_finish(ctx, &s, nullptr);
        ctx->head_unchecked = false;
        if (st != FQH_OK && st != FQH_E_CAPACITY) return st;
        const bool aligned = phi == 0 && prev_is_newline;  // the window begins with a record: nothing is in progress
        if (!aligned) {  // the record in progress must end inside the window
            if (s.n_records < 1) continue;
            uint64_t rs[2] = {0, 0};
            HIPCHK(ctx, hipMemcpyAsync(rs, ctx->d_misc, sizeof rs, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            offs[phi] = rs[1] - c.base_offset;
        }
        recs[phi] = s.n_records + (aligned ? 1 : 0);  // (counted alike: the unvalidated head record, or one for free)
        clean[phi] = s.parse_status == FQH_OK;
        if (clean[phi]) {
            ++n_ok;
            found = (int)phi;
        }
    }
    ctx->last_valid = false;
    ctx->spec_skip = keep_skip;
    ctx->spec_backoff = keep_backoff;
    if (n_ok > 1) return fail(ctx, FQH_E_ARG, "fqh_shard_align: more than one line phase validates (window too small)");
    if (n_ok == 0) {
        // The window holds a parse error.  Under the true phase the records in front of it still validate; under a wrong
        // one the first record after the head fails at once.  Take the phase that gets furthest, if it stands out: the
        // stream then reports the error where it is, and the ranks' phase check at the end guards the choice.
        for (uint32_t phi = 0; phi < 4; ++phi)
            if (found < 0 || recs[phi] > recs[found]) found = (int)phi;
        bool stands_out = true;
        for (uint32_t phi = 0; phi < 4; ++phi)
            if ((int)phi != found && recs[phi] + 1 >= recs[found]) stands_out = false;
        if (!stands_out || recs[found] < 3)
            return fail(ctx, FQH_E_HEADER, "fqh_shard_align: no line phase validates (a parse error at the window's start, or not FASTQ)");
    }
    off = offs[found];
    *phase = (uint32_t)found;
    *first_record_offset = off;
    return FQH_OK;
}

fqh_status fqh_rescan_launch(fqh_ctx *ctx, int is_final, const fqh_carry *in, uint64_t *d_rec_start,
                             uint64_t cap) {
    if (!ctx) return FQH_E_ARG;
    if (!ctx->last_valid) return fail(ctx, FQH_E_ARG, "no finished scan whose tile index could be reused");
    const uint8_t *buf = ctx->args.buf;
    const uint64_t len = ctx->args.len;
    return do_scan_launch(ctx, buf, len, is_final, in, d_rec_start, cap, true);
}


fqh_status fqh_carry_combine(const fqh_carry *prev, uint64_t len, uint64_t n_newlines,
                             uint64_t n_line_starts, const uint64_t back0[4], fqh_carry *next) {
    if (!next || !back0) return FQH_E_ARG;
    fqh_carry p = {};
    if (prev) p = *prev;
    fqh_carry n = {};
    n.base_offset = p.base_offset + len;
    n.nl_count = p.nl_count + n_newlines;
    // most recent line starts <= end of the shard: first the ones inside the shard (offsets 1..len),
    // then the one at shard offset 0 if the previous shard ended a line, then the previous carry's.
    int k = 0;
    for (; k < 4 && (uint64_t)k < n_line_starts; ++k) n.back[k] = back0[k];
    int j = 0;
    if (len == 0) {
        for (int i = 0; i < 4; ++i) n.back[i] = p.back[i];
    } else {
        for (; k < 4; ++k, ++j) {
            uint64_t v = p.back[j < 4 ? j : 3] + len;
            n.back[k] = v > n.base_offset ? n.base_offset : v;
        }
    }
    *next = n;
    return FQH_OK;
}

fqh_status fqh_index_records(fqh_ctx *ctx, fqh_idx_record *d_index, uint64_t cap) {
    if (!ctx || !d_index) return FQH_E_ARG;
    if (!ctx->last_valid || ctx->pending || ctx->stats_pending) return fail(ctx, FQH_E_ARG, "no finished scan to index");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const uint64_t n = std::min<uint64_t>(ctx->last_summary.n_records, cap);
    if (!n) return FQH_OK;
    fqh_status st = emit_index(ctx, d_index, n);
    if (st != FQH_OK) return st;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return FQH_OK;
}

}  // extern "C"
