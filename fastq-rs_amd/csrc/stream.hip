// stream.hip — fqh_stream_*: pinned-host ring + side-stream hipMemcpyAsync in front of the scan.
// Replaces the reference's Buffer (src/buffer.rs) and thread_reader (src/thread_reader.rs) for the
// GPU path: ingest of slot c+1 overlaps the scan of slot c and the caller's walk over slot c-1.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <new>
#include <vector>

#include "ctx.h"

struct fqh_stream {
    fqh_ctx *ctx = nullptr;
    uint32_t n_slots = 0, flags = 0;
    uint64_t slot_bytes = 0, reserve = 0;
    hipStream_t copy_stream = nullptr;
    struct Slot {
        uint8_t *h = nullptr;   // pinned: [reserve][slot_bytes] ([reserve] only in a ring of FQH_STREAM_EXTERNAL)
        uint8_t *d_base = nullptr, *d = nullptr;  // device: [reserve][slot_bytes + 16], d = d_base + reserve
        uint64_t h_bytes = 0, d_bytes = 0;
        const uint8_t *ext = nullptr;  // fqh_stream_submit_external: the caller's (registered) memory the slot's bytes came from
        uint64_t *d_rec = nullptr, *h_rec = nullptr;
        uint64_t rec_cap = 0;
        fqh_idx_record *d_idx = nullptr, *h_idx = nullptr;
        uint64_t idx_cap = 0;
        hipEvent_t copied = nullptr;
        uint64_t n_new = 0, lead = 0;
        int is_final = 0;
        bool launched = false;  // its scan is already enqueued (by the collect of the slot in front of it)
        bool fused = false;     // ... as a single pass (scan + histograms), commit held back
        hipEvent_t got = nullptr;  // its boundaries have arrived on the host
        hipEvent_t idle = nullptr; // everything the context's stream was given to do on the slot's device data has run
        bool idle_set = false;
        hipEvent_t tc0 = nullptr, tc1 = nullptr, ts0 = nullptr, ts1 = nullptr;  // FQH_STREAM_TIMING: its copy and its scan, begin / end
        int state = 0;  // 0 free, 1 acquired, 2 submitted, 3 collected (held by the caller)
    };
    std::vector<Slot> slots;
    uint64_t head = 0, sub = 0, col = 0;  // next slot to acquire / submit / collect (monotone counters)
    fqh_carry carry = {};
    uint64_t records_done = 0;
    bool ended = false;
    bool holds_exact = false;  // this stream keeps the context on the exact path (counted in fqh_ctx::exact_holds)
    // FQH_STREAM_TIMING: when each slot's copy (side stream) and scan (the context's stream) ran, in ms since t_base
    hipEvent_t t_base = nullptr;
    std::vector<float> iv_copy, iv_scan;  // begin, end, begin, end, ..
    // FQH_STREAM_STATS
    uint32_t lmax = 0;
    uint64_t *d_qual_hist = nullptr, *d_base_hist = nullptr, *d_scalars = nullptr;
    // fqh_stream_note_read: the host's reader may come back short (a pipe, a decompressor); "too long" is then judged by the
    // replay of the reference's Buffer under the noted read sizes instead of the closed form (csrc/replay.h)
    bool replay_on = false;
    fqh::BufferReplay replay;
};

static fqh_status grow_rec(fqh_stream *st, fqh_stream::Slot &s, uint64_t need) {
    fqh_ctx *ctx = st->ctx;
    if (s.rec_cap >= need) return FQH_OK;
    (void)hipFree(s.d_rec);
    if (s.h_rec) (void)hipHostFree(s.h_rec);
    s.d_rec = nullptr; s.h_rec = nullptr; s.rec_cap = 0;
    HIPCHK(ctx, hipMalloc((void **)&s.d_rec, need * sizeof(uint64_t)));
    HIPCHK(ctx, hipHostMalloc((void **)&s.h_rec, need * sizeof(uint64_t), hipHostMallocDefault));
    s.rec_cap = need;
    return FQH_OK;
}
static fqh_status grow_idx(fqh_stream *st, fqh_stream::Slot &s, uint64_t need) {
    fqh_ctx *ctx = st->ctx;
    if (s.idx_cap >= need) return FQH_OK;
    (void)hipFree(s.d_idx);
    if (s.h_idx) (void)hipHostFree(s.h_idx);
    s.d_idx = nullptr; s.h_idx = nullptr; s.idx_cap = 0;
    HIPCHK(ctx, hipMalloc((void **)&s.d_idx, need * sizeof(fqh_idx_record)));
    HIPCHK(ctx, hipHostMalloc((void **)&s.h_idx, need * sizeof(fqh_idx_record), hipHostMallocDefault));
    s.idx_cap = need;
    return FQH_OK;
}

void fqh_internal_free_parked(fqh_ctx *ctx) {
    if (!ctx || ctx->parked.empty()) return;
    (void)hipSetDevice(ctx->device);
    for (auto &k : ctx->parked) {
        if (k.h) (void)hipHostFree(k.h);
        (void)hipFree(k.d_base);
        (void)hipFree(k.d_rec);
        if (k.h_rec) (void)hipHostFree(k.h_rec);
        (void)hipFree(k.d_idx);
        if (k.h_idx) (void)hipHostFree(k.h_idx);
    }
    ctx->parked.clear();
}

extern "C" {

void fqh_stream_destroy(fqh_stream *st) {
    if (!st) return;
    (void)hipSetDevice(st->ctx->device);
    if (st->copy_stream) (void)hipStreamSynchronize(st->copy_stream);
    for (auto &s : st->slots)
        if (s.launched) {  // a scan enqueued ahead of its collect: end it, the context must not stay "pending"
            (void)fqh_internal_scan_finish(st->ctx, nullptr, nullptr);
            fqh_internal_fused_drop(st->ctx);
            s.launched = false;
        }
    (void)hipStreamSynchronize(st->ctx->stream);
    fqh_ctx *const ctx = st->ctx;
    // FQH_OPT_KEEP_RING: the buffers of ONE geometry stay with the context — the biggest seen (a gap of a few hundred bytes
    // opens a ring of 64 KiB slots next to the range's 255 MiB ones: those are not worth keeping)
    bool park = ctx->keep_ring && !st->slots.empty() && st->slots[0].h && st->slots[0].d_base;
    if (park && !ctx->parked.empty()) {
        const uint64_t have = ctx->parked[0].h_bytes + ctx->parked[0].d_bytes, mine = st->slots[0].h_bytes + st->slots[0].d_bytes;
        const bool same = ctx->parked[0].h_bytes == st->slots[0].h_bytes && ctx->parked[0].d_bytes == st->slots[0].d_bytes;
        if (!same && mine > have) fqh_internal_free_parked(ctx);
        else if (!same) park = false;
    }
    for (auto &s : st->slots) {
        if (park && s.h && s.d_base && ctx->parked.size() < 16) {
            fqh_ctx::ParkedSlot k;
            k.h = s.h; k.d_base = s.d_base; k.h_bytes = s.h_bytes; k.d_bytes = s.d_bytes;
            k.d_rec = s.d_rec; k.h_rec = s.h_rec; k.rec_cap = s.rec_cap;
            k.d_idx = s.d_idx; k.h_idx = s.h_idx; k.idx_cap = s.idx_cap;
            ctx->parked.push_back(k);
        } else {
            if (s.h) (void)hipHostFree(s.h);
            (void)hipFree(s.d_base);
            (void)hipFree(s.d_rec);
            if (s.h_rec) (void)hipHostFree(s.h_rec);
            (void)hipFree(s.d_idx);
            if (s.h_idx) (void)hipHostFree(s.h_idx);
        }
        if (s.copied) (void)hipEventDestroy(s.copied);
        if (s.got) (void)hipEventDestroy(s.got);
        if (s.idle) (void)hipEventDestroy(s.idle);
        for (hipEvent_t e : {s.tc0, s.tc1, s.ts0, s.ts1})
            if (e) (void)hipEventDestroy(e);
    }
    if (st->t_base) (void)hipEventDestroy(st->t_base);
    if (st->copy_stream) (void)hipStreamDestroy(st->copy_stream);
    if (st->holds_exact && st->ctx->exact_holds) --st->ctx->exact_holds;
    delete st;
}

fqh_status fqh_stream_create(fqh_ctx *ctx, uint64_t slot_bytes, uint32_t n_slots, uint32_t flags,
                             fqh_stream **out) {
    if (!ctx || !out || n_slots < 2 || slot_bytes < 4096 || (flags & ~(FQH_STREAM_INDEX | FQH_STREAM_STATS | FQH_STREAM_TIMING | FQH_STREAM_EXTERNAL)))
        return FQH_E_ARG;
    *out = nullptr;
    fqh_stream *st = new (std::nothrow) fqh_stream();
    if (!st) return FQH_E_DEVICE;
    st->ctx = ctx;
    st->n_slots = n_slots;
    st->flags = flags;
    st->slot_bytes = (slot_bytes + 15) & ~(uint64_t)15;
    // room for the partial trailing record of the previous slot: a record longer than the context's BUFSIZE is an
    // error in the reference, so two of them always suffice; without a limit (bufsize 0) records of up to 16 MiB
    // (or a slot, if that is less) are kept contiguous and anything longer is FQH_E_CAPACITY, not "too long"
    const uint64_t two = 2 * (uint64_t)FQH_BUFSIZE;
    const uint64_t want = ctx->bufsize ? 2 * ctx->bufsize : (st->slot_bytes < (16u << 20) ? st->slot_bytes : (uint64_t)(16u << 20));
    st->reserve = ((want > two ? want : two) + 15) & ~(uint64_t)15;
    st->slots.resize(n_slots);
    if (flags & FQH_STREAM_INDEX) {  // every chunk needs complete line lists: the context stays on the exact path while any
        st->holds_exact = true;      // such stream lives (a count, not a saved flag: streams may be destroyed in any order).
        ++ctx->exact_holds;          // (FQH_STREAM_STATS alone does not: its chunks take the single pass, k_scan_stats)
    }
    fqh_status rc = FQH_OK;
    do {
        if (hipSetDevice(ctx->device) != hipSuccess) { rc = FQH_E_DEVICE; break; }
        if (hipStreamCreateWithFlags(&st->copy_stream, hipStreamNonBlocking) != hipSuccess) { rc = FQH_E_DEVICE; break; }
        if ((flags & FQH_STREAM_TIMING) && (hipEventCreate(&st->t_base) != hipSuccess || hipEventRecord(st->t_base, ctx->stream) != hipSuccess)) { rc = FQH_E_DEVICE; break; }
        const uint64_t h_bytes = st->reserve + ((flags & FQH_STREAM_EXTERNAL) ? 0 : st->slot_bytes);
        const uint64_t d_bytes = st->reserve + st->slot_bytes + 16;
        for (auto &s : st->slots) {
            s.h_bytes = h_bytes;
            s.d_bytes = d_bytes;
            if (!ctx->parked.empty() && ctx->parked.back().h_bytes == h_bytes && ctx->parked.back().d_bytes == d_bytes) {
                const fqh_ctx::ParkedSlot k = ctx->parked.back();   // (FQH_OPT_KEEP_RING: a ring of this geometry lived before)
                ctx->parked.pop_back();
                s.h = k.h; s.d_base = k.d_base;
                s.d_rec = k.d_rec; s.h_rec = k.h_rec; s.rec_cap = k.rec_cap;
                s.d_idx = k.d_idx; s.h_idx = k.h_idx; s.idx_cap = k.idx_cap;
            } else if (hipHostMalloc((void **)&s.h, h_bytes, hipHostMallocDefault) != hipSuccess ||
                       hipMalloc((void **)&s.d_base, d_bytes) != hipSuccess) { rc = FQH_E_DEVICE; break; }
            if (hipEventCreateWithFlags(&s.copied, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&s.got, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&s.idle, hipEventDisableTiming) != hipSuccess) { rc = FQH_E_DEVICE; break; }
            if ((flags & FQH_STREAM_TIMING) && (hipEventCreate(&s.tc0) != hipSuccess || hipEventCreate(&s.tc1) != hipSuccess ||
                                                 hipEventCreate(&s.ts0) != hipSuccess || hipEventCreate(&s.ts1) != hipSuccess)) { rc = FQH_E_DEVICE; break; }
            s.d = s.d_base + st->reserve;  // reserve is a multiple of 16
            if (grow_rec(st, s, st->slot_bytes / 64 + 16) != FQH_OK) { rc = FQH_E_DEVICE; break; }
            if ((flags & FQH_STREAM_INDEX) && grow_idx(st, s, st->slot_bytes / 64 + 16) != FQH_OK) { rc = FQH_E_DEVICE; break; }
        }
    } while (0);
    if (rc != FQH_OK) {
        ctx->err = "fqh_stream_create: allocation failed";
        fqh_stream_destroy(st);
        return rc;
    }
    *out = st;
    return FQH_OK;
}

fqh_status fqh_stream_set_stats(fqh_stream *st, uint32_t lmax, uint64_t *d_qual_hist, uint64_t *d_base_hist,
                                uint64_t *d_scalars) {
    if (!st || !(st->flags & FQH_STREAM_STATS) || !lmax || !d_qual_hist || !d_base_hist || !d_scalars) return FQH_E_ARG;
    st->lmax = lmax;
    st->d_qual_hist = d_qual_hist;
    st->d_base_hist = d_base_hist;
    st->d_scalars = d_scalars;
    return FQH_OK;
}

fqh_status fqh_stream_acquire(fqh_stream *st, uint8_t **h_dst, uint64_t *cap) {
    if (!st || !h_dst || !cap) return FQH_E_ARG;
    if (st->flags & FQH_STREAM_EXTERNAL) return FQH_E_ARG;  // (its slots have no pinned data area)
    fqh_stream::Slot &s = st->slots[st->head % st->n_slots];
    if (st->head != st->sub || s.state != 0) return FQH_E_CAPACITY;  // previous acquire not submitted, or ring full
    s.state = 1;
    *h_dst = s.h + st->reserve;
    *cap = st->slot_bytes;
    ++st->head;
    return FQH_OK;
}

fqh_status fqh_stream_submit_external(fqh_stream *st, const uint8_t *h_src, uint64_t nbytes, int is_final) {
    if (!st || (!h_src && nbytes)) return FQH_E_ARG;
    fqh_stream::Slot &s = st->slots[st->head % st->n_slots];
    if (st->head != st->sub) return FQH_E_ARG;      // an acquired slot has not been submitted
    if (s.state != 0) return FQH_E_CAPACITY;        // ring full: collect / release first
    if (nbytes > st->slot_bytes) return FQH_E_ARG;
    s.state = 1;
    s.ext = h_src ? h_src : s.h + st->reserve;  // (nbytes == 0: nothing is read; the slot's own lead area stays in front)
    ++st->head;
    const fqh_status rc = fqh_stream_submit(st, nbytes, is_final);
    if (rc != FQH_OK) {  // (the slot goes back: nothing of it is in flight)
        s.state = 0;
        s.ext = nullptr;
        --st->head;
    }
    return rc;
}

fqh_status fqh_stream_submit(fqh_stream *st, uint64_t nbytes, int is_final) {
    if (!st) return FQH_E_ARG;
    if (st->sub + 1 != st->head) return FQH_E_ARG;  // nothing acquired
    fqh_stream::Slot &s = st->slots[st->sub % st->n_slots];
    if (s.state != 1 || nbytes > st->slot_bytes) return FQH_E_ARG;
    fqh_ctx *ctx = st->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const uint8_t *const src = s.ext ? s.ext : s.h + st->reserve;
    s.n_new = nbytes;
    s.is_final = is_final ? 1 : 0;
    // (the copy may not overtake what the context's stream still has to do on this slot's previous contents: commit kernels
    // that read its bytes, the move of its partial trailing record — enqueued by the collect that handed the slot back)
    if (s.idle_set) HIPCHK(ctx, hipStreamWaitEvent(st->copy_stream, s.idle, 0));
    if (s.tc0) HIPCHK(ctx, hipEventRecord(s.tc0, st->copy_stream));
    if (nbytes) HIPCHK(ctx, hipMemcpyAsync(s.d, src, nbytes, hipMemcpyHostToDevice, st->copy_stream));
    if (s.tc1) HIPCHK(ctx, hipEventRecord(s.tc1, st->copy_stream));
    HIPCHK(ctx, hipEventRecord(s.copied, st->copy_stream));
    s.state = 2;
    ++st->sub;
    return FQH_OK;
}

// Enqueues the scan of slot s under the carry `cy` (reuse: on the tile index of the attempt before, after its arrays grew).
// FQH_STREAM_STATS without FQH_STREAM_INDEX: as a single pass, scan + histograms in one read of the slot (k_scan_stats), with
// the commit held back until the host knows that every record of the slot is delivered.
static fqh_status launch_slot(fqh_stream *st, fqh_stream::Slot &s, const fqh_carry &cy, bool reuse) {
    fqh_ctx *ctx = st->ctx;
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, s.copied, 0));
    if (s.ts0) HIPCHK(ctx, hipEventRecord(s.ts0, ctx->stream));
    s.fused = false;
    struct NoLong {   // (the ring judges "too long" itself, on the boundaries it downloads: csrc/replay.h)
        fqh_ctx *c;
        explicit NoLong(fqh_ctx *x) : c(x) { c->no_long_rule = true; }
        ~NoLong() { c->no_long_rule = false; }
    } no_long(ctx);
    if (!reuse && (st->flags & FQH_STREAM_STATS) && !(st->flags & FQH_STREAM_INDEX) && st->lmax) {
        bool fused = false;
        fqh_status rc = fqh_internal_fused_launch(ctx, s.d, s.n_new, s.is_final, &cy, s.d_rec, s.rec_cap, st->lmax, st->d_qual_hist,
                                                  st->d_base_hist, st->d_scalars, s.lead, &fused);
        if (rc != FQH_OK) return rc;
        s.fused = fused;
    }
    if (!s.fused) {
        fqh_status rc = fqh_internal_scan_launch(ctx, s.d, s.n_new, s.is_final, &cy, s.d_rec, s.rec_cap, reuse);
        if (rc != FQH_OK) return rc;
    }
    if (s.ts1) HIPCHK(ctx, hipEventRecord(s.ts1, ctx->stream));
    s.launched = true;
    return FQH_OK;
}

fqh_status fqh_stream_collect(fqh_stream *st, fqh_chunk *out) {
    if (!st || !out) return FQH_E_ARG;
    if (st->col >= st->sub) return FQH_E_ARG;  // nothing submitted
    if (st->ended) return FQH_E_ARG;
    fqh_stream::Slot &s = st->slots[st->col % st->n_slots];
    if (s.state != 2) return FQH_E_ARG;
    // (a host that holds several chunks at once — fqh_stream_release_chunk — may still hold the chunk that lives in the NEXT slot
    // of the ring: this collect would write its partial trailing record in front of that slot's data, under the holder's eyes)
    if (!s.is_final && st->slots[(st->col + 1) % st->n_slots].state == 3) return FQH_E_AGAIN;
    fqh_ctx *ctx = st->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const bool want_stats = (st->flags & FQH_STREAM_STATS) && st->lmax;
    fqh_summary sum = {};
    fqh_carry cout = {};
    fqh_status rc;
    for (int attempt = 0;; ++attempt) {
        if (!s.launched) {
            rc = launch_slot(st, s, st->carry, attempt > 0);
            if (rc != FQH_OK) return rc;
        }
        s.launched = false;
        rc = fqh_internal_scan_finish(ctx, &sum, &cout);
        if (rc == FQH_E_CAPACITY && attempt == 0) {
            fqh_internal_fused_drop(ctx);  // (the histograms of this attempt are not committed: the rerun below counts them in two passes)
            s.fused = false;
            rc = grow_rec(st, s, sum.n_records + 16);
            if (rc != FQH_OK) return rc;
            continue;
        }
        if (rc != FQH_OK) return rc;
        break;
    }
    const uint64_t n = sum.n_records;
    // boundaries (and index) back to the host (n + 1 <= rec_cap: the scan returns FQH_E_CAPACITY otherwise, whatever
    // the parse status, and the loop above has grown the arrays and rescanned)
    if (n + 1 > s.rec_cap) return FQH_E_CAPACITY;
    HIPCHK(ctx, hipMemcpyAsync(s.h_rec, s.d_rec, (n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    if ((st->flags & FQH_STREAM_INDEX) && n) {
        rc = grow_idx(st, s, n);
        if (rc != FQH_OK) return rc;
        rc = fqh_internal_emit_index(ctx, s.d_idx, n);
        if (rc != FQH_OK) return rc;
        HIPCHK(ctx, hipMemcpyAsync(s.h_idx, s.d_idx, n * sizeof(fqh_idx_record), hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(ctx, hipEventRecord(s.got, ctx->stream));
    const uint64_t base_offset = st->carry.base_offset;
    const uint64_t known_end = base_offset + s.n_new;
    const uint64_t need0 = sum.parse_status != FQH_OK ? fqh_internal_last_need(ctx) : UINT64_MAX;
    // ---- before the host waits for the boundaries: what can already go to the GPU.  The reference's "record too long"
    // (src/lib.rs:278-283) can only bite a record of BUFSIZE - 15 bytes or more: if the slot holds none (and its partial
    // trailing record is shorter too), every record the scan found is delivered, and
    //   * the single pass's histograms are committed now,
    //   * the partial trailing record goes in front of the next slot's device data,
    //   * the NEXT slot's scan is enqueued under this one's carry-out: it runs while the host walks this slot.
    const bool single = s.fused && fqh_internal_fused_owed(ctx);
    const bool no_long = !ctx->bufsize || (sum.max_record_len + 15 < ctx->bufsize && sum.tail_len + 15 < ctx->bufsize);
    const bool clean = sum.parse_status == FQH_OK && no_long && sum.tail_len <= st->reserve;
    bool stats_done = !want_stats;
    if (n == 0 && !stats_done) {  // nothing is delivered with this slot: nothing to count
        fqh_internal_fused_drop(ctx);
        stats_done = true;
    }
    bool dev_tail_done = false, idle_early = false;
    if (clean) {
        if (single) {
            fqh_internal_fused_commit(ctx);
            stats_done = true;
        }
        fqh_stream::Slot &nx = st->slots[(st->col + 1) % st->n_slots];
        if (!s.is_final && sum.tail_len && want_stats) {  // device twin of the tail move (may reach into s's own lead)
            HIPCHK(ctx, hipMemcpyAsync(nx.d - sum.tail_len, s.d + s.n_new - sum.tail_len, sum.tail_len, hipMemcpyDeviceToDevice, ctx->stream));
            dev_tail_done = true;
        }
        if (stats_done && !s.is_final && st->col + 1 < st->sub && nx.state == 2 && !nx.launched) {
            nx.lead = sum.tail_len;
            // everything the context's stream has to do on THIS slot's device data is enqueued (commit, tail move): the event its
            // next copy waits for goes in front of the next slot's scan, not behind it — with two slots the refill of this one
            // would otherwise wait for that scan instead of running under it
            if (!want_stats || dev_tail_done || !sum.tail_len) {
                HIPCHK(ctx, hipEventRecord(s.idle, ctx->stream));
                idle_early = true;
            }
            rc = launch_slot(st, nx, cout, false);
            if (rc != FQH_OK) return rc;
        }
    }
    HIPCHK(ctx, hipEventSynchronize(s.got));
    if (s.tc0 && st->iv_copy.size() < (1u << 22)) {  // (all four events lie behind: the scan was finished above)
        float a = 0, b = 0, c0 = 0, c1 = 0;
        if (hipEventElapsedTime(&a, st->t_base, s.tc0) == hipSuccess && hipEventElapsedTime(&b, st->t_base, s.tc1) == hipSuccess &&
            hipEventElapsedTime(&c0, st->t_base, s.ts0) == hipSuccess && hipEventElapsedTime(&c1, st->t_base, s.ts1) == hipSuccess) {
            st->iv_copy.push_back(a);
            st->iv_copy.push_back(b);
            st->iv_scan.push_back(c0);
            st->iv_scan.push_back(c1);
        }
        (void)hipGetLastError();
    }

    fqh_chunk c = {};
    c.parse_status = sum.parse_status;
    c.is_final = s.is_final;
    c.n_records = n;
    c.base_offset = base_offset;
    c.data_len = s.n_new;
    // (a slot whose bytes came from the caller's own memory: that memory is the chunk's host view; the beginning of the record
    // in progress at its start lies at the END of the chunk before it, not in front of h_data — d_data has it in front)
    c.lead_len = s.ext ? 0 : s.lead;
    c.h_data = s.ext ? s.ext : s.h + st->reserve;
    c.h_index = (st->flags & FQH_STREAM_INDEX) ? s.h_idx : nullptr;
    c.h_rec_start = s.h_rec;
    c.d_data = s.d;
    c.d_rec_start = s.d_rec;
    c.err_record = sum.err_record;
    c.err_offset = sum.err_offset;
    c.err_need = need0;

    // "Fastq record is too long" (src/lib.rs:278-283): the reference sees the record that starts at file offset p through a window
    // of BUFSIZE - (p & 15) bytes, whatever came before it (csrc/replay.h, fqh::TooLong: the closed form of the Buffer's
    // arithmetic) — the boundaries are file offsets (fqh_stream_set_origin for a stream that does not begin the file), so every
    // slot is judged by itself, the record in progress at its end included
    uint64_t which = 0;
    const bool bad_here = sum.parse_status != FQH_OK;
    const uint64_t need = bad_here ? need0 : fqh::TooLong::NO_BAD;
    bool too_long;
    bool late = false;       // the replay's verdict names a record of an EARLIER chunk (see fqh_stream_note_read in the header)
    uint64_t late_k = 0, late_off = 0;
    if (st->replay_on) {   // (a reader that comes back short: the reference's reads are replayed, fqh_stream_note_read)
        uint64_t k = 0;
        too_long = st->replay.step(s.h_rec, st->records_done, n, known_end, s.is_final || bad_here, need, &k);
        if (too_long && k < st->records_done) {
            // (defensive: the verdict on a record falls before its last byte is read, so a record is judged with the chunk it ends
            // in and this is not known to be reachable — tests/test_gpu_stream.py searches for it; if it ever is, name the record)
            late = true;
            late_k = k;
            late_off = st->replay.pend.empty() ? 0 : st->replay.pend[0];   // (step() has dropped the boundaries in front of k)
        }
        which = k >= st->records_done ? std::min<uint64_t>(k - st->records_done, n) : 0;
    } else {
        too_long = fqh::TooLong::first(ctx->bufsize, s.h_rec, n, known_end - s.h_rec[n], need, &which);
    }
    if (too_long) {
        c.parse_status = FQH_E_TOO_LONG;
        c.err_record = late ? late_k : st->records_done + which;
        c.n_records = which;
        c.err_offset = late ? late_off : s.h_rec[which];
    }
    // histograms of the records this chunk delivers (the one in progress at its start included: its
    // beginning sits in front of the slot's device data)
    if (!stats_done && c.n_records) {
        if (single && c.n_records == n) {
            fqh_internal_fused_commit(ctx);   // (no scan of a later slot is in flight: the context still describes this one)
        } else {
            fqh_internal_fused_drop(ctx);
            ctx->trust_index = true;   // (the slot's bytes are the ring's own: nobody has written them since the scan above)
            rc = fqh_internal_stats_launch(ctx, s.d, s.n_new, s.is_final, &st->carry, st->lmax, st->d_qual_hist,
                                           st->d_base_hist, st->d_scalars, s.lead, c.n_records);
            ctx->trust_index = false;
            if (rc != FQH_OK) return rc;
            rc = fqh_stats_finish(ctx, nullptr, nullptr);
            if (rc != FQH_OK) return rc;
        }
    } else if (!stats_done) {
        fqh_internal_fused_drop(ctx);
    }
    // the partial trailing record goes in front of the next slot's data
    const uint64_t tail = known_end - s.h_rec[n];
    if (c.parse_status == FQH_OK && !s.is_final) {
        if (tail > st->reserve) {
            if (!ctx->bufsize || tail + 15 < ctx->bufsize) {  // the reference would accept it; the ring cannot hold it
                ctx->err = "fqh_stream: a record in progress is longer than the lead area of the ring (bufsize 0: 16 MiB or one slot)";
                return FQH_E_CAPACITY;
            }
            c.parse_status = FQH_E_TOO_LONG;  // cannot be kept contiguous; the reference rejects it as well
            c.err_record = st->records_done + n;
            c.err_offset = s.h_rec[n];
            c.err_need = tail;
        } else {
            fqh_stream::Slot &nx = st->slots[(st->col + 1) % st->n_slots];
            if (tail) {
                uint8_t *const dst = nx.h + st->reserve - tail;
                if (!s.ext || tail <= s.n_new) {
                    memcpy(dst, (s.ext ? s.ext : s.h + st->reserve) + s.n_new - tail, tail);
                } else {  // no record ended in a slot of the caller's memory: the record in progress began in the slot's own lead area
                    const uint64_t old = tail - s.n_new;
                    memcpy(dst, s.h + st->reserve - old, old);
                    memcpy(dst + old, s.ext, s.n_new);
                }
            }
            if (tail && want_stats && !dev_tail_done) {  // device twin of the same move (may reach into s's own lead)
                HIPCHK(ctx, hipMemcpyAsync(nx.d - tail, s.d + s.n_new - tail, tail, hipMemcpyDeviceToDevice, ctx->stream));
                HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            }
            nx.lead = tail;
        }
    }
    if (!idle_early) HIPCHK(ctx, hipEventRecord(s.idle, ctx->stream));
    s.idle_set = true;
    if (c.parse_status != FQH_OK || s.is_final) st->ended = true;
    st->records_done += n;
    st->carry = cout;
    s.state = 3;
    ++st->col;
    *out = c;
    return FQH_OK;
}

fqh_status fqh_stream_timing(fqh_stream *st, fqh_stream_times *out) {
    if (!st || !out) return FQH_E_ARG;
    *out = fqh_stream_times{};
    const std::vector<float> &c = st->iv_copy, &k = st->iv_scan;
    out->n_slots = c.size() / 2;
    if (c.empty()) return FQH_OK;
    // both lists are in time order and their intervals do not overlap among themselves (one stream each)
    double busy_c = 0, busy_k = 0, both = 0;
    for (size_t i = 0; i + 1 < c.size(); i += 2) busy_c += c[i + 1] - c[i];
    for (size_t i = 0; i + 1 < k.size(); i += 2) busy_k += k[i + 1] - k[i];
    for (size_t i = 0, j = 0; i + 1 < c.size() && j + 1 < k.size();) {
        const float lo = std::max(c[i], k[j]), hi = std::min(c[i + 1], k[j + 1]);
        if (hi > lo) both += hi - lo;
        if (c[i + 1] < k[j + 1]) i += 2; else j += 2;
    }
    out->wall_ms = std::max(c.back(), k.back()) - std::min(c.front(), k.front());
    out->copy_busy_ms = busy_c;
    out->scan_busy_ms = busy_k;
    out->both_busy_ms = both;
    return FQH_OK;
}

fqh_status fqh_stream_set_origin(fqh_stream *st, uint64_t file_offset) {
    if (!st || st->head || st->sub || st->col) return FQH_E_ARG;  // before the first acquire
    st->carry = fqh_carry{};
    st->carry.base_offset = file_offset;
    return FQH_OK;
}

fqh_status fqh_stream_note_read(fqh_stream *st, uint64_t got, uint64_t asked) {
    if (!st || got > asked) return FQH_E_ARG;
    // (the replay asks for up to BUFSIZE bytes at a time: a host whose first ask of a slot could be less cannot say what the
    // reference's reader would have got; and the BUFSIZE the replay was started with must still be the context's)
    if (st->ctx->bufsize && st->slot_bytes < st->ctx->bufsize) return FQH_E_ARG;
    if (st->replay_on && st->replay.B != st->ctx->bufsize) return FQH_E_ARG;
    if (!st->replay_on) {
        // the replay walks the reference's Buffer from the first byte of the file: the notes must begin with the first slot
        if (st->sub || st->col || st->carry.base_offset) return FQH_E_ARG;
        st->replay.reset(st->ctx->bufsize);
        st->replay_on = true;
    }
    st->replay.note_read(got, asked);
    return FQH_OK;
}

fqh_status fqh_stream_carry(fqh_stream *st, fqh_carry *out) {
    if (!st || !out) return FQH_E_ARG;
    *out = st->carry;
    return FQH_OK;
}

fqh_status fqh_stream_release(fqh_stream *st) {
    if (!st || st->col == 0) return FQH_E_ARG;
    fqh_stream::Slot &s = st->slots[(st->col - 1) % st->n_slots];
    if (s.state != 3) return FQH_E_ARG;
    s.state = 0;
    s.lead = 0;
    s.launched = false;
    s.ext = nullptr;
    return FQH_OK;
}

fqh_status fqh_stream_release_chunk(fqh_stream *st, const fqh_chunk *c) {
    if (!st || !c || !c->d_data) return FQH_E_ARG;
    for (auto &s : st->slots)
        if (s.d == c->d_data) {
            if (s.state != 3) return FQH_E_ARG;
            s.state = 0;
            s.lead = 0;
            s.launched = false;
            s.ext = nullptr;
            return FQH_OK;
        }
    return FQH_E_ARG;
}

fqh_status fqh_host_register(fqh_ctx *ctx, void *h_ptr, uint64_t bytes) {
    if (!ctx || !h_ptr || !bytes) return FQH_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipHostRegister(h_ptr, bytes, hipHostRegisterDefault));
    return FQH_OK;
}

fqh_status fqh_host_unregister(fqh_ctx *ctx, void *h_ptr) {
    if (!ctx || !h_ptr) return FQH_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipHostUnregister(h_ptr));
    return FQH_OK;
}

}  // extern "C"
