// ctx.h — the library-internal definition of fqh_ctx, shared by context.hip, scan_dispatch.hip, stats_dispatch.hip and stream.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "fqh_internal.h"
#include "replay.h"

namespace fqh {
void launch_index(hipStream_t, const uint8_t *, uint64_t, uint16_t *, uint32_t, uint32_t *, uint16_t *, uint64_t, DevOut *, int, bool);
void launch_emit_fast(hipStream_t, const ScanArgs &, DevOut *, int);
void launch_finalize_fast(hipStream_t, const ScanArgs &, DevOut *);
void launch_prefix(hipStream_t, uint32_t *, const uint16_t *, uint32_t *, uint64_t *, uint64_t, uint64_t);
void launch_emit(hipStream_t, const ScanArgs &, DevOut *, int);
void launch_finalize(hipStream_t, const ScanArgs &, DevOut *);
size_t stats_oct_scratch_bytes(uint32_t, int);
hipError_t launch_stats_oct(hipStream_t, StatsArgs, int);
hipError_t launch_stats_long(hipStream_t, const uint8_t *, uint64_t, uint64_t, const fqh_idx_record *, uint64_t, uint32_t, uint32_t, uint32_t *,
                             uint64_t, unsigned long long *, unsigned long long *, unsigned long long *, int, uint32_t *);
size_t stats_long_scratch_bytes(uint64_t, uint64_t, uint32_t, int);
void launch_shard_words(hipStream_t, const DevOut *, const DevOut *, uint64_t, uint64_t *);
void launch_carry_fold(hipStream_t, const uint64_t *, int, int, DevCarry *, DevCarry *, DevOut *);
void launch_shard_counts(hipStream_t, const DevOut *, const DevOut *, const DevCarry *, uint64_t *);
bool scan_stats_supports(uint32_t lmax, uint32_t hint, bool mostly_long);
uint32_t scan_stats_blocks(uint64_t n_tiles, int n_cu);
size_t scan_stats_scratch_bytes(int n_cu);
hipError_t launch_scan_stats(hipStream_t, FusedArgs, int);
void launch_stats_commit(hipStream_t, const DevOut *, const FusedArgs &, uint32_t, unsigned long long *, unsigned long long *,
                         unsigned long long *);
hipError_t prepare_stats_declined(uint32_t rows);
void launch_stats_declined(hipStream_t, const DevOut *, const FusedArgs &, unsigned long long *, unsigned long long *, unsigned long long *);
uint32_t scan_stats_nsl(uint32_t rows);
uint32_t scan_stats_rows(uint32_t lmax, uint32_t hint);
void launch_peek_lines(hipStream_t, const uint8_t *, uint64_t, uint32_t, unsigned long long *);
uint32_t peek_windows(uint64_t len);
void launch_stats_head(hipStream_t, const StatsArgs &, const uint64_t back[4]);
void launch_stats_edge(hipStream_t, const DevOut *, const uint8_t *, uint64_t, uint64_t, int, uint32_t, unsigned long long *,
                       unsigned long long *, unsigned long long *);
void launch_record_flags(hipStream_t, const uint8_t *, uint64_t, uint64_t, const fqh_idx_record *, uint64_t, uint8_t *);
uint64_t gather_blocks(uint64_t n);
void launch_gather(hipStream_t, const uint8_t *, uint64_t, uint64_t, const fqh_idx_record *, uint64_t, const uint8_t *, uint32_t,
                   uint32_t, unsigned long long *, unsigned long long *, unsigned long long *, uint8_t *, uint64_t);
void launch_synth(hipStream_t, uint8_t *, uint64_t, uint64_t, uint64_t);
void launch_read_ceiling(hipStream_t, const uint8_t *, uint64_t, uint64_t *, int);
}  // namespace fqh

using namespace fqh;

struct fqh_ctx {
    int device = 0;
    int n_cu = 256;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    uint64_t bufsize = FQH_BUFSIZE;
    std::string err;

    // workspace (grow-only)
    uint16_t *list = nullptr;
    size_t list_elems = 0;
    uint32_t list_cap = LIST_CAP_DEFAULT;
    uint32_t *tile_count = nullptr, *tile_prefix = nullptr;
    unsigned long long *gather_ws = nullptr;  // fqh_gather_records: block byte sums, block record sums, totals
    uint64_t gather_ws_blocks = 0;
    uint16_t *fast_rs = nullptr;  // fast path: per tile two 128-byte lines (record starts | edges, count, alignment)
    uint64_t *block_prefix = nullptr;
    size_t tiles_cap = 0;
    DevOut *d_out = nullptr;      // [0] the scan's, [1] scratch for index-only emits
    DevOut *h_out = nullptr;      // pinned
    DevOut *h_init = nullptr;     // pinned reset image
    uint64_t *d_misc = nullptr;   // 32 u64 of scratch
    bool placed = false;             // the line buffer in use has been through place_fast_rs
    int place_tries = 0;             // candidates of the fast path's per-tile lines the first big scan may allocate and time (FQH_OPT_PLACE_TRIES)
    int spin_wait_us = 0;            // FQH_OPT_SPIN_WAIT: poll the stream this long in fqh_*_finish before sleeping on it
    float place_ms[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // what the candidates measured (0: not tried); [8] = the chosen one's, [9] = without stores
    int place_n = 0;                 // candidates the search tried (0: no search ran)
    // Adaptive choice of the fast path's line buffer (FQH_OPT_ADAPT_LINES): whether the index kernel runs at 2.65 or
    // at 2.83 ms per 16 GiB is a property of the PAIR (input allocation, line-buffer allocation).  A context therefore keeps up
    // to two line buffers and learns, per big input it sees again (address, length, kernel), which one that input runs faster
    // with — from the HIP-event time of the real scans, one buffer per call: no extra launches, no blocking search.
    struct LinesAdapt {
        const uint8_t *buf = nullptr;
        uint64_t len = 0;
        bool fused = false;
        int state = 0;            // 0 new (the first buffer is being measured), 1 measured: the next call takes the alternate, 3 settled
        int choice = 0;           // settled: which of fr[] this input takes
        int tries = 0;            // alternates that measured like the primary and were given back
        int seen = 0, seen_alt = 0;  // measurements of the first buffer / of the alternate being tried (two each, the faster counts)
        float ms[2] = {0, 0};
        uint64_t stamp = 0;
    } adapt[4];
    uint16_t *fr[2] = {nullptr, nullptr};  // fr[0]: the buffer the workspace was allocated with; fr[1]: the alternate (same size), or NULL
    int adapt_max = 3;            // alternates tried per input (0: off)
    uint16_t *fr_rejects[8] = {};  // alternates that measured like fr[0], held until the input they were tried for is settled
    int n_rejects = 0;
    int adapt_entry = -1, adapt_used = 0;  // the launch in flight: which entry is being measured, with which buffer
    uint64_t adapt_clock = 0;
    uint16_t *list_dummy = nullptr;  // 1 KiB: where k_index_fast's list-area writes go while the context has no line lists
    DevCarry *d_carry = nullptr;  // device-side shard protocol: the folded carry, and its pinned twin the host reads at finish
    DevCarry *h_carry = nullptr;
    bool dev_carry = false;       // the launch in flight took its carry from d_carry
    fqh_idx_record *idx = nullptr;
    size_t idx_cap = 0;
    fqh_idx_record *scan_idx = nullptr;  // set around a scan launch: its emit step also writes the record index there
    uint64_t scan_idx_cap = 0;
    bool idx_emitted = false;     // the last scan's emit step wrote the index of its records to ctx->idx
    uint64_t *tmp_rec = nullptr;
    size_t tmp_rec_cap = 0;
    uint32_t *stats_scratch = nullptr;
    size_t stats_scratch_bytes = 0;

    hipEvent_t ev[8] = {};
    fqh_timing timing = {};

    // the scan in flight / last finished
    bool pending = false;
    bool last_valid = false;
    ScanArgs args = {};
    fqh_carry carry_in = {};
    bool whole_file = false;
    bool no_long_rule = false;      // set around a launch whose "too long" rule is someone else's (the ring's slots) or has no file offset to go by
    bool launch_long_rule = true;   // ... as captured by the launch in flight
    bool skip_emit = false;  // shard prescan: only the byte scan, the prefix and the chunk-end summary
    bool head_unchecked = false;  // fqh_shard_align: the record in progress at the chunk start is not validated
    // fast path: prove validity with a quarter of the list traffic; any doubt -> exact rerun
    bool spec_enabled = true;   // false: exact path only (callers that need full line lists, FQH_SPEC=0)
    uint32_t exact_holds = 0;   // live fqh_streams that need complete line lists for every chunk: no fast path while > 0
    uint32_t rows_hint = 0;     // the longest sequence / quality line this context knows of in the kind of input it is given (0: nothing yet):
                                // what the single pass sizes its rows by (scan_stats_rows)
    const uint8_t *hint_buf = nullptr;   // ... and the input that belief is about: buffer, length, file offset of the last statistics call
    uint64_t hint_len = 0, hint_base = 0;
    bool hint_valid = false;
    uint32_t f_rows = 0;        // ... the rows of the single pass in flight
    bool lines_long = false;    // ... and whether most of its lines are longer than the single pass takes (511 bytes): kilobase reads
    uint32_t fused_skip = 0, fused_backoff = 0;  // statistics calls left on the two-pass route after a single pass that had to be given up
                                                 // (reads longer than the histogram's rows, more dirty lines than the dump area holds): 1, 2, 4 .. 64
    uint32_t spec_skip = 0;     // scans left on the exact path after the fast path failed ...
    uint32_t spec_backoff = 0;  // ... 1, 2, 4 .. 64 of them, doubling with every failure in a row
    bool reuse_index = false;   // FQH_OPT_REUSE_INDEX: fqh_stats* may count over the last scan's tile index (the caller vouches for the bytes)
    bool trust_index = false;   // ... the library's own scan-then-count sequence (the ring) is running
    bool index_full = true;     // the tile index in the workspace holds complete line lists
    bool fast_needs_list = false;  // an input of this context had tiles denser than the fast path's two lines: keep the line lists allocated
    bool dout_clean = false;    // d_out[0]'s accumulators were reset by the last finalize kernel (no init copy needed)
    bool used_spec = false;     // the scan in flight runs the fast path
    fqh_summary last_summary = {};
    fqh_carry last_carry_out = {};
    // stats in flight
    bool stats_pending = false;
    fqh_status stats_cap_st = FQH_OK;  // two-pass fqh_scan_stats: the scan's FQH_E_CAPACITY, reported by the finish
    // single-pass scan + histograms (k_scan_stats): the request of the launch in flight, and the side arrays the
    // kernel's 64-bit counters go to until k_stats_commit adds them to the caller's
    bool fused = false;           // the scan being enqueued / in flight counts as well
    uint32_t f_lmax = 0;
    uint64_t *f_qual = nullptr, *f_base = nullptr, *f_scalars = nullptr;
    uint64_t f_lead = 0;          // bytes of valid device memory in front of the buffer (the record in progress at the chunk start)
    bool f_defer_commit = false;  // the stream's: the commit kernels wait until the host has replayed the reference's Buffer
    bool f_commit_owed = false;   // a deferred commit of the last finished single-pass launch has not been enqueued yet
    FusedArgs f_args = {};        // the launch's kernel arguments (for the deferred commit)
    unsigned long long *side = nullptr;   // FQH_NSCALARS totals of the launch in flight
    uint32_t *decl_b = nullptr;   // the single pass's dump area (batches it would not count) and list of long lines, counted
    uint64_t *decl_l = nullptr;   // exactly by k_stats_declined (fqh_internal.h: FusedArgs); one slot per 512 KiB of input + 1024
    uint32_t decl_cap = 0;
    size_t decl_b_bytes = 0;
    int stats_route = 0;          // fqh_last_stats_route: how the last finished statistics call counted
    // FQH_OPT_KEEP_RING: the pinned / device buffers of destroyed rings, parked for the next fqh_stream_create of the same
    // geometry (pinning hundreds of MiB takes tens of ms: a host that opens one ring per file, or per fqh_shard_stream_run,
    // pays it once per context instead)
    struct ParkedSlot {
        uint8_t *h = nullptr, *d_base = nullptr;
        uint64_t h_bytes = 0, d_bytes = 0;
        uint64_t *d_rec = nullptr, *h_rec = nullptr;
        uint64_t rec_cap = 0;
        fqh_idx_record *d_idx = nullptr, *h_idx = nullptr;
        uint64_t idx_cap = 0;
    };
    std::vector<ParkedSlot> parked;
    bool keep_ring = false;
    bool fused_enabled = true;    // FQH_FUSED=0: histograms always as a second pass over a full index
};

#define HIPCHK(ctx, call)                                                                      \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                    \
            return FQH_E_DEVICE;                                                               \
        }                                                                                      \
    } while (0)


// internal entry points shared between translation units
fqh_status fqh_internal_scan_launch(fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, int is_final,
                                    const fqh_carry *in, uint64_t *d_rec_start, uint64_t cap, bool reuse_index);
fqh_status fqh_internal_scan_finish(fqh_ctx *ctx, fqh_summary *out, fqh_carry *carry_out);
fqh_status fqh_internal_emit_index(fqh_ctx *ctx, fqh_idx_record *dst, uint64_t cap);
fqh_status fqh_internal_stats_launch(fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, int is_final, const fqh_carry *in,
                                     uint32_t lmax, uint64_t *d_qual_hist, uint64_t *d_base_hist, uint64_t *d_scalars,
                                     uint64_t lead_len, uint64_t n_limit);
fqh_status fqh_internal_fused_launch(fqh_ctx *ctx, const uint8_t *d_buf, uint64_t len, int is_final, const fqh_carry *in,
                                     uint64_t *d_rec_start, uint64_t cap, uint32_t lmax, uint64_t *d_qual_hist,
                                     uint64_t *d_base_hist, uint64_t *d_scalars, uint64_t lead_len, bool *fused);
bool fqh_internal_fused_owed(const fqh_ctx *ctx);
void fqh_internal_fused_commit(fqh_ctx *ctx);
void fqh_internal_fused_drop(fqh_ctx *ctx);
// frees the ring buffers a context has parked (FQH_OPT_KEEP_RING)
void fqh_internal_free_parked(fqh_ctx *ctx);
// error visibility of the failing record of the last finished scan (BufferReplay::step's `need`)
uint64_t fqh_internal_last_need(const fqh_ctx *ctx);
