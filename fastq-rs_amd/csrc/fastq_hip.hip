// fastq_hip.hip — host side of libfastq_hip.so: the C ABI of include/fastq_hip.h.
//
// Owns the per-context workspace in HBM (line-start lists, tile counts/prefixes), enqueues the
// kernels of scan_kernels.hip / stats_kernels.hip on one HIP stream, reads back one small struct
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

static fqh_status fail(fqh_ctx *ctx, fqh_status s, const char *msg) {
    if (ctx) ctx->err = msg;
    return s;
}

// Bytes of one line buffer: first lines (+ 64 tiles: k_emit_fast loads whole rounds of tiles without clamping), then the second
// lines (+ 64 again).  Tuning builds keep 4 MiB of slack behind it so that an experiment can shift the lines inside the
// allocation (FQH_TUNE_LINES_OFFSET, tools/exp_lines_offset.py).
static size_t lines_bytes(size_t n_tiles) {
#ifdef FQH_TUNING
    return (2 * n_tiles + 128) * 64 * sizeof(uint16_t) + (4u << 20);
#else
    return (2 * n_tiles + 128) * 64 * sizeof(uint16_t);
#endif
}
static uint16_t *lines_in_use(fqh_ctx *ctx) {
#ifdef FQH_TUNING
    if (const char *e = getenv("FQH_TUNE_LINES_OFFSET")) return ctx->fast_rs + (((size_t)atoll(e) & ~(size_t)127) & ((4u << 20) - 1)) / sizeof(uint16_t);
#endif
    return ctx->fast_rs;
}

// Every line buffer of the fast path the context holds — the one in use (fast_rs), the adaptive choice's two (fr[]) and the
// alternates held back until an input is settled (fr_rejects[]) — freed once each: the same allocation may sit in several of them.
static void free_line_buffers(fqh_ctx *ctx) {
    uint16_t *all[3 + 8];
    int n = 0;
    auto add = [&](uint16_t *p) {
        for (int i = 0; i < n; ++i)
            if (all[i] == p) return;
        if (p) all[n++] = p;
    };
    add(ctx->fast_rs);
    add(ctx->fr[0]);
    add(ctx->fr[1]);
    for (int i = 0; i < ctx->n_rejects; ++i) add(ctx->fr_rejects[i]);
    for (int i = 0; i < n; ++i) (void)hipFree(all[i]);
    ctx->fast_rs = ctx->fr[0] = ctx->fr[1] = nullptr;
    ctx->n_rejects = 0;
    for (auto &e : ctx->adapt) e = fqh_ctx::LinesAdapt{};
}

extern "C" {

int fqh_abi_version(void) { return FQH_ABI_VERSION; }

const char *fqh_strerror(fqh_status s) {
    switch (s) {
    case FQH_OK: return "ok";
    case FQH_E_HEADER: return "Fastq headers must start with '@'";
    case FQH_E_SEP: return "Sequence and quality not separated by +";
    case FQH_E_LEN_MISMATCH: return "Sequence and quality length mismatch";
    case FQH_E_TRUNCATED: return "Possibly truncated input file";
    case FQH_E_TOO_LONG: return "Fastq record is too long";
    case FQH_E_IO: return "i/o error";
    case FQH_E_DEVICE: return "HIP device error";
    case FQH_E_ARG: return "invalid argument";
    case FQH_E_CAPACITY: return "output capacity too small";
    case FQH_E_AGAIN: return "a shard left the fast path: use the host recipe";
    }
    return "unknown";
}

static thread_local std::string g_create_err = "no context";   // (fqh_last_error(NULL): why THIS thread's last fqh_create failed)
const char *fqh_last_error(fqh_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

fqh_status fqh_create(int device, fqh_ctx **out) {
    if (!out) return FQH_E_ARG;
    *out = nullptr;
    int n = 0;
    hipError_t he = hipGetDeviceCount(&n);
    if (he != hipSuccess || n <= 0 || device < 0 || device >= n) {
        g_create_err = std::string("hipGetDeviceCount: ") + hipGetErrorString(he) + ", devices=" + std::to_string(n);
        return FQH_E_DEVICE;
    }
    fqh_ctx *ctx = new (std::nothrow) fqh_ctx();
    if (!ctx) return FQH_E_DEVICE;
    ctx->device = device;
    fqh_status st = FQH_OK;
    g_create_err = "HIP resource creation failed";
    do {
        if (hipSetDevice(device) != hipSuccess) { st = FQH_E_DEVICE; break; }
        int cu = 0;
        if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cu > 0)
            ctx->n_cu = cu;
        // A BLOCKING stream: a caller that works on the legacy null stream (hipMemsetAsync(.., 0), torch's default stream) and
        // never calls fqh_set_stream gets the ordering it expects — its fills are done before the kernels here read or
        // overwrite the memory, and its reads see what they wrote.  (A non-blocking stream raced with a torch.zeros() of the
        // offsets array in the tests: the fill ran late and wiped part of the result.)  Callers with streams of their own
        // pass one (fqh_set_stream) and order their work on it.
        if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamDefault) != hipSuccess) { st = FQH_E_DEVICE; break; }
        ctx->stream = ctx->own_stream;
        if (hipMalloc((void **)&ctx->d_out, 2 * sizeof(DevOut)) != hipSuccess) { st = FQH_E_DEVICE; break; }
        if (hipMalloc((void **)&ctx->d_misc, 64) != hipSuccess) { st = FQH_E_DEVICE; break; }
        if (hipMalloc((void **)&ctx->list_dummy, 1024) != hipSuccess) { st = FQH_E_DEVICE; break; }
        if (hipHostMalloc((void **)&ctx->h_out, sizeof(DevOut), hipHostMallocDefault) != hipSuccess) { st = FQH_E_DEVICE; break; }
        if (hipHostMalloc((void **)&ctx->h_init, sizeof(DevOut), hipHostMallocDefault) != hipSuccess) { st = FQH_E_DEVICE; break; }
        memset(ctx->h_init, 0, sizeof(DevOut));
        ctx->h_init->min_key = NOKEY;
        ctx->h_init->first_long = NOKEY;
        for (auto &e : ctx->ev)
            if (hipEventCreate(&e) != hipSuccess) { st = FQH_E_DEVICE; break; }
    } while (0);
    if (st != FQH_OK) {
        fqh_destroy(ctx);
        return st;
    }
    if (const char *e = getenv("FQH_SPEC")) ctx->spec_enabled = atoi(e) != 0;    // knob: 0 = exact path only
    if (const char *e = getenv("FQH_FUSED")) ctx->fused_enabled = atoi(e) != 0;  // knob: 0 = histograms as a second pass
    *out = ctx;
    return FQH_OK;
}

void fqh_destroy(fqh_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (auto &e : ctx->ev)
        if (e) (void)hipEventDestroy(e);
    fqh_internal_free_parked(ctx);
    (void)hipFree(ctx->list);
    (void)hipFree(ctx->tile_count);
    (void)hipFree(ctx->tile_prefix);
    free_line_buffers(ctx);
    (void)hipFree(ctx->block_prefix);
    (void)hipFree(ctx->d_out);
    (void)hipFree(ctx->d_misc);
    (void)hipFree(ctx->list_dummy);
    (void)hipFree(ctx->d_carry);
    if (ctx->h_carry) (void)hipHostFree(ctx->h_carry);
    (void)hipFree(ctx->idx);
    (void)hipFree(ctx->tmp_rec);
    (void)hipFree(ctx->stats_scratch);
    (void)hipFree(ctx->gather_ws);
    (void)hipFree(ctx->side);
    (void)hipFree(ctx->decl_b);
    (void)hipFree(ctx->decl_l);
    if (ctx->h_out) (void)hipHostFree(ctx->h_out);
    if (ctx->h_init) (void)hipHostFree(ctx->h_init);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

fqh_status fqh_set_stream(fqh_ctx *ctx, void *hip_stream) {
    if (!ctx) return FQH_E_ARG;
    if (ctx->pending || ctx->stats_pending) return fail(ctx, FQH_E_ARG, "a launch is pending");
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return FQH_OK;
}

fqh_status fqh_set_bufsize(fqh_ctx *ctx, uint64_t bufsize) {
    if (!ctx) return FQH_E_ARG;
    if (bufsize && (bufsize < 32 || bufsize % 16)) return fail(ctx, FQH_E_ARG, "bufsize must be 0 or a multiple of 16 >= 32");
    ctx->bufsize = bufsize;
    ctx->last_valid = false;
    return FQH_OK;
}

}  // extern "C"

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
static void enqueue_fused_commit(fqh_ctx *ctx) {
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

static fqh_status ensure_full_index(fqh_ctx *ctx);

// The cached tile index describes the BYTES of the last scanned buffer: every writer of this library that touches
// them drops it.  (Writes the library cannot see — the caller's own kernels, a reused allocator block — need
// fqh_invalidate; see include/fastq_hip.h.)
static void drop_index_if_overlaps(fqh_ctx *ctx, const void *d_dst, uint64_t bytes) {
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

static fqh_status do_scan_launch(fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, int is_final,
                                 const fqh_carry *in, uint64_t *d_rec_start, uint64_t cap,
                                 bool reuse_index = false) {
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

static fqh_status do_scan_finish(fqh_ctx *ctx, fqh_summary *out, fqh_carry *carry_out) {
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
static bool same_scan(const fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, int is_final, const fqh_carry *in) {
    if (!ctx->last_valid || !(ctx->reuse_index || ctx->trust_index)) return false;
    fqh_carry c = {};
    if (in) c = *in;
    return ctx->args.buf == d_buf && ctx->args.len == len && ctx->args.is_final == (is_final ? 1 : 0) &&
           memcmp(&c, &ctx->carry_in, sizeof c) == 0;
}

// index-only emit of the last scan into `dst` (device), local records [0, n)
// (re)build complete line lists for the buffer of the last scan (after a fast-path scan)
static fqh_status ensure_full_index(fqh_ctx *ctx) {
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

static fqh_status emit_index(fqh_ctx *ctx, fqh_idx_record *dst, uint64_t cap) {
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

int fqh_last_scan_fast(fqh_ctx *ctx) { return ctx && ctx->used_spec ? 1 : 0; }
int fqh_last_stats_route(fqh_ctx *ctx) { return ctx ? ctx->stats_route : 0; }
fqh_status fqh_set_option(fqh_ctx *ctx, int option, int value) {
    if (!ctx) return FQH_E_ARG;
    if (ctx->pending || ctx->stats_pending) return fail(ctx, FQH_E_ARG, "a launch is pending");
    switch (option) {
    case FQH_OPT_FAST_PATH:
        ctx->spec_enabled = value != 0;
        ctx->spec_skip = ctx->spec_backoff = 0;
        return FQH_OK;
    case FQH_OPT_SINGLE_PASS:
        ctx->fused_enabled = value != 0;
        ctx->fused_skip = ctx->fused_backoff = 0;
        ctx->rows_hint = 0;
        ctx->lines_long = false;
        ctx->hint_valid = false;
        return FQH_OK;
    case FQH_OPT_PLACE_TRIES:
        ctx->place_tries = value < 0 ? 0 : value > 8 ? 8 : value;
        return FQH_OK;
    case FQH_OPT_REUSE_INDEX:
        ctx->reuse_index = value != 0;
        return FQH_OK;
    case FQH_OPT_ADAPT_LINES:
        ctx->adapt_max = value < 0 ? 0 : value > 8 ? 8 : value;
        return FQH_OK;
    case FQH_OPT_SPIN_WAIT:
        ctx->spin_wait_us = value < 0 ? 0 : value > 1000000 ? 1000000 : value;
        return FQH_OK;
    case FQH_OPT_KEEP_RING:
        ctx->keep_ring = value != 0;
        if (!ctx->keep_ring) fqh_internal_free_parked(ctx);
        return FQH_OK;
    case FQH_OPT_OWN_STREAM_NONBLOCKING: {
        // the context's own stream again, blocking (ordered against the legacy null stream: the safe default) or not (no
        // coupling with the process's null-stream work: the caller orders what it hands in with events, as for any stream)
        if (ctx->pending || ctx->stats_pending) return fail(ctx, FQH_E_ARG, "a launch is pending");
        if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, FQH_E_DEVICE, "hipSetDevice");
        hipStream_t fresh = nullptr;
        if (hipStreamCreateWithFlags(&fresh, value ? hipStreamNonBlocking : hipStreamDefault) != hipSuccess) return fail(ctx, FQH_E_DEVICE, "hipStreamCreateWithFlags");
        (void)hipStreamSynchronize(ctx->own_stream);
        const bool in_use = ctx->stream == ctx->own_stream;
        (void)hipStreamDestroy(ctx->own_stream);
        ctx->own_stream = fresh;
        if (in_use) ctx->stream = fresh;
        return FQH_OK;
    }
    }
    return fail(ctx, FQH_E_ARG, "unknown option");
}
fqh_status fqh_placement(fqh_ctx *ctx, int *n_candidates, float ms[10]) {
    if (!ctx || !n_candidates || !ms) return FQH_E_ARG;
    *n_candidates = ctx->place_n;
    for (int i = 0; i < 10; ++i) ms[i] = ctx->place_ms[i];
    return FQH_OK;
}
fqh_status fqh_line_buffers(fqh_ctx *ctx, int *n_alive, int *n_unsettled, uint64_t *bytes) {
    if (!ctx) return FQH_E_ARG;
    const uint16_t *all[3 + 8];
    int n = 0;
    auto add = [&](const uint16_t *p) {
        for (int i = 0; i < n; ++i)
            if (all[i] == p) return;
        if (p) all[n++] = p;
    };
    add(ctx->fast_rs);
    add(ctx->fr[0]);
    add(ctx->fr[1]);
    for (int i = 0; i < ctx->n_rejects; ++i) add(ctx->fr_rejects[i]);
    int open_inputs = 0;
    for (const auto &e : ctx->adapt) open_inputs += e.buf && e.state != 3 ? 1 : 0;
    if (n_alive) *n_alive = n;
    if (n_unsettled) *n_unsettled = ctx->adapt_max > 0 ? open_inputs : 0;
    if (bytes) *bytes = (uint64_t)n * lines_bytes(ctx->tiles_cap);
    return FQH_OK;
}
fqh_status fqh_invalidate(fqh_ctx *ctx) {
    if (!ctx) return FQH_E_ARG;
    ctx->last_valid = false;
    return FQH_OK;
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
    unsigned long long peek[3] = {0, 0, 0};
    launch_peek_lines(ctx->stream, d_buf, len, in ? (uint32_t)(in->nl_count & 3u) : 0u, (unsigned long long *)ctx->d_misc);
    HIPCHK(ctx, hipMemcpyAsync(peek, ctx->d_misc, sizeof peek, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t seen = (uint32_t)std::min<unsigned long long>(std::max<unsigned long long>(peek[0], 1), 0x7FFFFFFFull);
    // the rows go up with what a look finds, down with what the scans find (resolve()); whether MOST lines are too long for the
    // pass is what the look says (64 KiB of kilobase reads: every other line)
    if (seen > ctx->rows_hint || !ctx->rows_hint) ctx->rows_hint = seen;
    ctx->lines_long = peek[2] * 4 > peek[1] || (peek[1] == 0 && len >= 65536);   // (no newline in 64 KiB)
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

fqh_status fqh_last_timing(fqh_ctx *ctx, fqh_timing *out) {
    if (!ctx || !out) return FQH_E_ARG;
    *out = ctx->timing;
    return FQH_OK;
}

fqh_status fqh_synth_fill(fqh_ctx *ctx, uint8_t *d_out, uint64_t byte_off, uint64_t len, uint64_t seed) {
    if (!ctx || (len && !d_out)) return FQH_E_ARG;
    if (len && ((uintptr_t)d_out & 15)) return fail(ctx, FQH_E_ARG, "d_out must be 16-byte aligned");
    drop_index_if_overlaps(ctx, d_out, len);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    launch_synth(ctx->stream, d_out, byte_off, len, seed);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return FQH_OK;
}

fqh_status fqh_read_ceiling(fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, uint64_t *checksum, float *ms) {
    if (!ctx || (len && !d_buf)) return FQH_E_ARG;
    if (len && ((uintptr_t)d_buf & 15)) return fail(ctx, FQH_E_ARG, "d_buf must be 16-byte aligned");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    HIPCHK(ctx, hipMemsetAsync(ctx->d_misc, 0, 64, s));
    HIPCHK(ctx, hipEventRecord(ctx->ev[7], s));
    launch_read_ceiling(s, d_buf, len, ctx->d_misc, ctx->n_cu);
    HIPCHK(ctx, hipEventRecord(ctx->ev[6], s));
    HIPCHK(ctx, hipGetLastError());
    uint64_t sum = 0;
    HIPCHK(ctx, hipMemcpyAsync(&sum, ctx->d_misc, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    float t = 0;
    HIPCHK(ctx, hipEventElapsedTime(&t, ctx->ev[7], ctx->ev[6]));
    if (checksum) *checksum = sum;
    if (ms) *ms = t;
    return FQH_OK;
}

fqh_status fqh_dev_alloc(fqh_ctx *ctx, uint64_t bytes, void **d_ptr) {
    if (!ctx || !d_ptr) return FQH_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMalloc(d_ptr, bytes ? bytes : 16));
    return FQH_OK;
}
fqh_status fqh_dev_free(fqh_ctx *ctx, void *d_ptr) {
    if (!ctx) return FQH_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipFree(d_ptr));
    return FQH_OK;
}
fqh_status fqh_memcpy_h2d(fqh_ctx *ctx, void *d_dst, const void *h_src, uint64_t bytes) {
    if (!ctx) return FQH_E_ARG;
    drop_index_if_overlaps(ctx, d_dst, bytes);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return FQH_OK;
}
fqh_status fqh_memcpy_d2h(fqh_ctx *ctx, void *h_dst, const void *d_src, uint64_t bytes) {
    if (!ctx) return FQH_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return FQH_OK;
}
fqh_status fqh_memset(fqh_ctx *ctx, void *d_dst, int value, uint64_t bytes) {
    if (!ctx) return FQH_E_ARG;
    drop_index_if_overlaps(ctx, d_dst, bytes);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemsetAsync(d_dst, value, bytes, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return FQH_OK;
}

}  // extern "C"
