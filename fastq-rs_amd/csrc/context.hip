// context.hip — libfastq_hip.so, host side: contexts, options, timing, the synthetic generator and the device-memory helpers of
// include/fastq_hip.h.  (The record scan is scan_dispatch.hip, the statistics calls stats_dispatch.hip; the kernels are
// scan_kernels.hip / fused_kernels.hip / stats_kernels.hip / filter_kernels.hip.)
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

fqh_status fail(fqh_ctx *ctx, fqh_status s, const char *msg) {
    if (ctx) ctx->err = msg;
    return s;
}

// Bytes of one line buffer: first lines (+ 64 tiles: k_emit_fast loads whole rounds of tiles without clamping), then the second
// lines (+ 64 again).  Tuning builds keep 4 MiB of slack behind it so that an experiment can shift the lines inside the
// allocation (FQH_TUNE_LINES_OFFSET, tools/exp_lines_offset.py).
size_t lines_bytes(size_t n_tiles) {
#ifdef FQH_TUNING
    return (2 * n_tiles + 128) * 64 * sizeof(uint16_t) + (4u << 20);
#else
    return (2 * n_tiles + 128) * 64 * sizeof(uint16_t);
#endif
}
uint16_t *lines_in_use(fqh_ctx *ctx) {
#ifdef FQH_TUNING
    if (const char *e = getenv("FQH_TUNE_LINES_OFFSET")) return ctx->fast_rs + (((size_t)atoll(e) & ~(size_t)127) & ((4u << 20) - 1)) / sizeof(uint16_t);
#endif
    return ctx->fast_rs;
}

// Every line buffer of the fast path the context holds — the one in use (fast_rs), the adaptive choice's two (fr[]) and the
// alternates held back until an input is settled (fr_rejects[]) — freed once each: the same allocation may sit in several of them.
void free_line_buffers(fqh_ctx *ctx) {
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
        if (hipMalloc((void **)&ctx->d_misc, 256) != hipSuccess) { st = FQH_E_DEVICE; break; }
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

extern "C" {

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
