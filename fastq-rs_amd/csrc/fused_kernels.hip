// fused_kernels.hip — k_scan_stats: record scan AND per-position histograms in ONE read of the input
// (DESIGN.md §5b).  The reference touches every record once: Parser::each hands it to the closure that
// reads seq()/qual() (src/lib.rs:226-237, src/records.rs:83-90).  Here the byte scan of the fast path
// (k_index_fast, scan_kernels.hip) and the eight-lanes-per-line count of k_stats_oct (stats_dev.h) run in
// the same wavefront on the same LDS image of the data.
//
// What bounds it is the vector ALU (a wave64 VALU operation occupies its SIMD for four cycles: 4.8e9 of them
// per 16 GiB made the first version take 11 ms), then the LDS atomic unit; HBM comes third.  Hence:
//   * one block per CU owns the CU's LDS: the bank-scheduled histogram (stats_dev.h) and, behind it, 6 160 B per
//     wavefront: [512 B tail of the previous group | 4 KiB group | 512 B of the bytes after the span | 16 B |
//     256 record starts of the tile | 4 + 251 + 1 line-start entries];
//   * a wavefront takes SPANS of four 16 KiB tiles round-robin and walks them 4 KiB at a time like k_index_fast
//     (16-byte non-temporal loads a group ahead, SWAR newline masks, ballot prefix).  The LDS image is LINEAR; the
//     lane-contiguous read-back is conflict-free because lane l reads its four 16-byte chunks in the order
//     (i + l / 4) % 4 and rotates its 64-bit newline mask back (4 VALU per group);
//   * the group's line starts are staged in LDS; windows of five entries are checked under the four alignments
//     exactly as in k_index_fast.  The span's first group must single out one alignment: the phase everything in
//     the span is counted under; every tile must confirm it, and k_emit_fast checks each tile's against the true
//     global line index.  Any doubt sets spec_fail and nothing of this pass is used;
//   * every line that ENDS in a group is counted from LDS (the tail keeps the 512 bytes in front of the group, so
//     a line that straddles groups or tiles is contiguous): one lane per ENTRY first fetches the class bits of the
//     entry's window and the byte in front of the newline in one LDS round trip, then works out the line's start and
//     length (trim_winline: one '\r'); eight lines make a batch, a lane's dword of each step is one ds_read2_b32 with an
//     immediate offset and one v_alignbyte, then pass 1 / pass 2 of the bank schedule: one v_perm_b32 + one
//     ds_sub_u32 per byte.  A batch that is not full at the end of a group stays in registers and is filled up by
//     the next group's lines (batches are 97 % full instead of 77 %);
//   * the span's last line ends in another wavefront's span: the 512 bytes after the span are read as well and
//     the line is closed there;
//   * no exact path in here, and a count the kernel cannot make is NOT a doubt about the parse: a batch of eight lines with
//     a byte outside ACGTN / '!'..'`' is not counted but DUMPED — the lanes' line words and raw dwords, 1.5 KiB — and a
//     sequence / quality line longer than the histogram's rows is LISTED (offset, length); k_stats_declined counts both with
//     the plain per-byte statement behind k_stats_commit.  A line longer than the kept tail, or a full dump area, sets
//     DevOut::stats_declined: nothing of the pass is committed and the caller counts in a second pass (the scan's result
//     stands).  Only what the kernel cannot VALIDATE — more than 251 line starts in 4 KiB, an alignment that does not
//     stand out — marks the span bad and sets spec_fail (the caller reruns the exact path);
//   * per-block partial histograms and the totals go to scratch; k_stats_commit adds them to the caller's arrays
//     only if the scan's finalize kernel found no reason to doubt the fast path.
//
// Algorithmic bytes: len per launch — the only read of the input for offsets, validation and histograms.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "scan_dev.h"
#include "stats_dev.h"

namespace fqh {

#ifndef FQH_FZ_W5
#define FQH_FZ_W5 16
#endif
#ifndef FQH_WIDE_GUARD
#define FQH_WIDE_GUARD 0   // 1: the eight-step wide instance stops at the last step a line of the batch reaches (measured: slower, see fz_steps)
#endif
#ifdef FQH_TUNING
#define FZ_DBG(bit) ((z.dbg & (bit)) != 0)
#else
#define FZ_DBG(bit) false
#endif
constexpr uint32_t FZ_WAVES_MAX = 16;              // wavefronts per block: 16 with up to 160 rows, 12 with 256 (LDS)
#ifndef FQH_FZ_SPAN
#define FQH_FZ_SPAN 4
#endif
constexpr uint32_t FZ_SPAN = FQH_FZ_SPAN;                     // tiles a wavefront walks in one go
constexpr uint32_t FZ_GROUP = 4096;                 // bytes per group: 64 contiguous bytes per lane
constexpr uint32_t FZ_TAIL = 512;                   // bytes of the previous group kept in front of the group
constexpr uint32_t FZ_POST = 512;                   // bytes after the span, for the span's last line
constexpr uint32_t FZ_DATA = FZ_TAIL + FZ_GROUP + FZ_POST;
constexpr uint32_t FZ_GLIST = 251;                  // line starts per group that can be staged (+ one virtual entry)
constexpr uint32_t FZ_RS = 256;                    // record starts of a tile staged in LDS (4 x 251 entries / 4, rounded)
constexpr uint32_t FZ_WAVE_BYTES = FZ_DATA + 16 + FZ_RS * 2 + (4 + FZ_GLIST + 1) * 2;  // 6160
constexpr uint32_t FZ_SLACK = 1024;                 // a batch reads up to 32 NSL + 32 bytes past a line's start (NSL 16: 544)
// The WIDE instances (rows 161 .. 511: MiSeq 2 x 300, merged pairs).  Two things change against the instances
// of eight lanes per line:
//   * SIXTEEN lanes walk a line, 64 columns per step, four lines per batch: a lane keeps one word per step, so three to six
//     steps (192 .. 384 columns) fit the 128 registers of sixteen wavefronts per CU and eight steps (511 columns) the 168 of
//     twelve.  (Eight lanes per line and sixteen
//     steps of 32 — measured — need 246 registers, eight wavefronts, and ran the 300-column file at 1.55 TB/s, below the
//     two-read route.)  Lane (line slot g, dword m) adds byte j = k ^ (g & 1) at its k-th atomic: row 64 u + 4 m + j, slot
//     m + 16 j — the two line slots of a lane group differ in j & 1, so its 32 lanes sit on the 32 banks (m + 16 j) % 32
//     whatever the bins are;
//   * 512 rows x 64 quality bins of 32-bit counters are 128 KiB — no room for the wavefronts' areas — so two rows share a word,
//     16-bit counters: step u counts in half u & 1 of its word, a row block (u >> 1) holds 128 columns: 16 KiB of quality
//     histogram per 128 columns.  A counter may not pass 65 535 between two flushes: a span holds at most
// 4 tiles x 4 groups x 251 line starts / 4 = 1 004 records (denser input marks the span bad), so the block's wavefronts walk
// fz_epoch spans each, meet at a barrier, add the 16-bit halves to the block's 32-bit rows in scratch, clear the LDS and go on.
__host__ __device__ constexpr uint32_t fz_epoch(uint32_t waves) { return 65535u / (1004u * waves); }
static_assert(FZ_WAVE_BYTES % 16 == 0, "wave areas are read with 16-byte accesses");

typedef uint32_t fz_u32x2 __attribute__((ext_vector_type(2), aligned(4)));

// A kernel argument that only cold paths use (dumps, listings, the list area, the epilogue), re-read from the kernarg segment
// where it is used: held in scalar registers from the prologue on, the seventeen of them are spilled to vector lanes around
// the loops (tools/isa.sh: sgpr_spill_count).
template <class T>
__device__ __forceinline__ T fz_karg(uint32_t off) {
    const __attribute__((address_space(4))) char *p = (const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));  // (opaque: the load stays where the argument is used)
    return *reinterpret_cast<const __attribute__((address_space(4))) T *>(p + off);
}
#define FZ_KARG(f) fz_karg<decltype(FusedArgs::f)>((uint32_t)offsetof(FusedArgs, f))

// spans of FZ_SPAN tiles; a last partial tile of at most one group belongs to the span in front of it
__host__ __device__ __forceinline__ uint32_t fz_spans(uint64_t n_tiles, uint64_t len) {
    const uint64_t tail = len & (WT_BYTES - 1);
    const uint64_t own = (n_tiles > 1 && tail != 0 && tail <= FZ_GROUP) ? n_tiles - 1 : n_tiles;
    return (uint32_t)((own + FZ_SPAN - 1) / FZ_SPAN);
}

// A line as one lane knows it: bits 0-8 its length (columns), bit 12 "there is a line", bits 16-28 the LDS position y of
// its first byte in the wave's data area.  The low 16 bits are the shape key.
constexpr uint32_t FZ_P_ACT = 0x1000u;

struct FzLane {              // what a lane needs to read and count lines; constant over the kernel
    uint32_t wm4;            // LDS address of the wave's data area + 4 * m
    uint32_t m, g8, g16;     // the lane's dword of its line (lane % 8; WIDE: lane % 16), its line slot in the batch (lane / 8; WIDE: lane / 16), 16 * slot
    SoLane c;                // the bank schedule's selectors and slot offsets
};

template <uint32_t NSL>
struct FzBatch {             // eight lines in flight: this lane's dword of each step of its line
    uint32_t P;
    uint32_t w[NSL];
};

// What a lane derives from the shape of its line and its place in the group of eight; kept across batches and worked
// out again only when a line of another shape turns up.  Two modes.  Uniform (every line of the batch that ends in a
// partial dword has it at the same step `tus`, the rule with reads of one length): that dword is counted in step tus
// along with everything else — cm[tus] is its byte mask for the check, tv[] the values of its four atomics.  Ragged:
// cm[] covers whole dwords only and the partial dwords get their own pass (tb, tv, tu).
template <uint32_t NSL>
struct FzShape {
    uint32_t key;            // low 16 bits of the P it was derived from
    uint32_t cm[NSL];        // check mask of this lane's dword at step u; also the value of its atomics (0 / ~0) in all steps but tus
    uint32_t tv[4];          // what the k-th atomic ADDS (1: byte k ^ (g & 3) counts) at step tus / in the pass of the partial dwords
    uint32_t tb, tu;         // ragged mode: byte mask of the lane's partial dword (0: none) and its step
    uint32_t mode;           // wave-uniform: bits 0-7 tus (0xFF: ragged), bit 8: some line ends in a partial dword, bits 16-19: steps to issue
};
template <uint32_t NSL, bool WIDE>
__device__ __forceinline__ void fz_shape(FzShape<NSL> &S, uint32_t P, uint32_t m) {
    constexpr uint32_t CPS = WIDE ? 64u : 32u;   // columns per step
    S.key = P & 0xFFFFu;
    const uint32_t len = P & 0x1FFu;
    const uint32_t nfull4 = len & ~3u, nbt = len & 3u;
    const uint32_t tu = nfull4 / CPS, mt = (nfull4 >> 2) & (CPS / 4u - 1u), g3 = WIDE ? (__lane_id() >> 4) & 1u : (__lane_id() >> 3) & 3u;
    const unsigned long long tl = __ballot(nbt != 0);
    uint32_t tus = 0xFFu;
    if (tl) {
        const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)tu, (int)(__ffsll((long long)tl) - 1));
        if (__ballot(nbt != 0 && tu != t0) == 0) tus = t0;
    } else {
        tus = NSL;  // no partial dword anywhere: no step is special
    }
    const bool uni = tus != 0xFFu;
    const int tt = (int)nfull4 - (int)(4u * m);
    const uint32_t pmask = (nbt && m == mt) ? (1u << (8u * nbt)) - 1u : 0u;  // this lane's partial dword (at step tu)
#pragma unroll
    for (uint32_t u = 0; u < NSL; ++u)
        S.cm[u] = tt > (int)(CPS * u) ? 0xFFFFFFFFu : (uni && tu == u) ? pmask : 0u;
    const bool whole_at_tus = uni && tus < NSL && tt > (int)(CPS * tus);
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k)
        S.tv[k] = (whole_at_tus || (nbt && m == mt && (k ^ g3) < nbt)) ? 1u : 0u;
    S.tb = uni ? 0u : pmask;
    S.tu = nbt ? tu : 0u;
    // WIDE: bits 16-19 the steps this batch has to issue (the last one any of its lines reaches, partial dwords included)
    uint32_t nst = NSL;
    if (WIDE && NSL == 8 && FQH_WIDE_GUARD) {
        nst = 0;
#pragma unroll
        for (uint32_t u = 0; u < NSL; ++u)
            if (__ballot((int)len > (int)(CPS * u)) != 0) nst = u + 1;
    }
    S.mode = tus | (tl ? 0x100u : 0u) | (nst << 16);
}

// four atomics of one step: ds_sub_u32 of v_k at the address byte k of the bins gives, `off` in the instruction's immediate
template <bool IS_SEQ>
__device__ __forceinline__ void fz_sub4_at(const uint32_t off, const SoLane &c, uint32_t pb, uint32_t v0, uint32_t v1, uint32_t v2,
                                           uint32_t v3) {
    (void)__hip_atomic_fetch_sub((so_lds_u32 *)(uintptr_t)(__builtin_amdgcn_perm(c.slots, pb, c.sel[0]) + off), v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    (void)__hip_atomic_fetch_sub((so_lds_u32 *)(uintptr_t)(__builtin_amdgcn_perm(c.slots, pb, c.sel[1]) + off), v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    (void)__hip_atomic_fetch_sub((so_lds_u32 *)(uintptr_t)(__builtin_amdgcn_perm(c.slots, pb, c.sel[2]) + off), v2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    (void)__hip_atomic_fetch_sub((so_lds_u32 *)(uintptr_t)(__builtin_amdgcn_perm(c.slots, pb, c.sel[3]) + off), v3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Count one batch: pass 1 checks every byte the batch counts, pass 2 adds them — one v_perm_b32 and one ds_sub_u32 per
// byte (stats_dev.h).  No exact path: a batch with a byte outside the alphabet / window is dumped, not counted (fz_dump).
// Sequence dwords stay RAW in the batch (the bins are worked out again in pass 2: one v_and per step) so that a dump holds
// the bytes the file holds; quality dwords are rebased in place, which a dump undoes.
// A batch that holds a byte outside the alphabet / window: its lanes' line words (bit 31: quality lines) and raw dwords go to
// the dump area, 64 x (1 + NSL) words per batch; k_stats_declined counts them.  No room left: nothing of this pass is used.
// Has this pass been given up already (DevOut::stats_declined: nothing of it will be committed)?  Asked before every atomic that
// lists or dumps something: a pass over reads that are ALL longer than the histogram's rows (the caller's lmax is its own
// choice) otherwise queues millions of read-modify-writes of one counter — 25 ms per 4 GiB of 300-base reads with lmax = 150,
// against 3 ms for the same pass with its results thrown away.
__device__ __forceinline__ bool fz_given_up(const FusedArgs &z) {
    return __hip_atomic_load(&FZ_KARG(out)->stats_declined, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}
template <bool IS_SEQ, uint32_t NSL>
__device__ __forceinline__ void fz_dump(const FzBatch<NSL> &B, const FusedArgs &z) {
    if (fz_given_up(z)) return;
    uint32_t slot = 0;
    if (__lane_id() == 0) slot = (uint32_t)atomicAdd(&FZ_KARG(out)->decl_batches, 1ull);
    slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
    if (slot >= FZ_KARG(decl_cap)) {
        if (__lane_id() == 0) atomicAdd(&FZ_KARG(out)->stats_declined, 1ull);
        return;
    }
    uint32_t *dst = FZ_KARG(decl_b) + (uint64_t)slot * ((1u + NSL) * 64u) + __lane_id();
    dst[0] = (B.P & 0x7FFFFFFFu) | (IS_SEQ ? 0u : 0x80000000u);
#pragma unroll
    for (uint32_t u = 0; u < NSL; ++u) dst[64u * (1u + u)] = IS_SEQ ? B.w[u] : B.w[u] + 0x21212121u;  // (quality dwords were rebased in place)
}

// Steps 0 .. n - 1 of a batch, each a call of f(FzStep<u>).  Every instance issues all its steps, unrolled (n is the experiments').
template <uint32_t U>
struct FzStep { static constexpr uint32_t value = U; };
template <uint32_t U, uint32_t N, class F>
__device__ __forceinline__ void fz_steps_all(F &f) {
    if constexpr (U < N) {
        f(FzStep<U>{});
        fz_steps_all<U + 1, N>(f);
    }
}
// Measured and NOT taken, all in the eight-step wide instance (tools/build_variant.sh, cold fqh_stats of 8 GiB): stopping at the
// last step some line of the batch reaches — a wave-uniform compare and branch per step (FQH_WIDE_GUARD): 450 / 500 / 511 bp
// 2 192 / 2 260 / 2 269 GB/s against 2 357 / 2 424 / 2 442 with every step issued; one jump into a fall-through switch of steps
// instead of those compares (FQH_WIDE_SWITCH): 3 % below the compares; reading only the steps the longest line so far reaches
// (FQH_WIDE_NRD): 10 % below; switch and bounded reads together: 23 % below.  Less work, slower code: the join points cost the
// waits more than the skipped steps save.
#ifndef FQH_WIDE_SWITCH
#define FQH_WIDE_SWITCH 0
#endif
#ifndef FQH_WIDE_NRD
#define FQH_WIDE_NRD 0
#endif
template <uint32_t U, uint32_t N, class F>
__device__ __forceinline__ void fz_steps_guard(uint32_t n, F &f) {
    if constexpr (U < N) {
        if (U < n) f(FzStep<U>{});
        fz_steps_guard<U + 1, N>(n, f);
    }
}
template <uint32_t NSL, bool WIDE, class F>
__device__ __forceinline__ void fz_steps(uint32_t n, F &&f) {
    if constexpr (!WIDE || NSL < 8 || !FQH_WIDE_GUARD) {   // (every instance issues all its steps)
        fz_steps_all<0, NSL>(f);
    } else if constexpr (!FQH_WIDE_SWITCH) {
        fz_steps_guard<0, NSL>(n, f);
    } else {
        static_assert(NSL == 8, "the wide instance has eight steps");
        switch (n) {
        default: f(FzStep<7>{}); [[fallthrough]];
        case 7: f(FzStep<6>{}); [[fallthrough]];
        case 6: f(FzStep<5>{}); [[fallthrough]];
        case 5: f(FzStep<4>{}); [[fallthrough]];
        case 4: f(FzStep<3>{}); [[fallthrough]];
        case 3: f(FzStep<2>{}); [[fallthrough]];
        case 2: f(FzStep<1>{}); [[fallthrough]];
        case 1: f(FzStep<0>{}); [[fallthrough]];
        case 0: break;
        }
    }
}
// byte offset (the instruction's immediate) of step u's rows: region, row block, half of the 256-byte bin row
template <bool WIDE>
__host__ __device__ constexpr uint32_t fz_off(uint32_t region, uint32_t rb, uint32_t u) {
    return WIDE ? region + (u >> 1) * rb : region + (u & 1u) * 128u + (u >> 1) * rb;
}
__device__ __forceinline__ uint32_t so_groups16(unsigned long long lanes) {  // 16-lane groups with a lane set
    lanes |= lanes >> 8;
    lanes |= lanes >> 4;
    lanes |= lanes >> 2;
    lanes |= lanes >> 1;
    return (uint32_t)__builtin_popcountll(lanes & 0x0001000100010001ull);
}
template <bool IS_SEQ, uint32_t NSL, bool WIDE>
__device__ __forceinline__ void fz_count(FzBatch<NSL> &B, FzShape<NSL> &S, const FzLane &L, SoTotals &T, bool &bad, const FusedArgs &z) {
    (void)bad;  // (not written any more: a batch the kernel will not count is dumped.  The parameter stays: without it the register
                // allocator spills 116 instead of 21 vector registers around the tile loop — tools/isa.sh)
    const SoLane &c = L.c;
    constexpr bool PACK = WIDE;   // (the wide instance is the packed one)
    if (__ballot((B.P & 0xFFFFu) != S.key) != 0) fz_shape<NSL, WIDE>(S, B.P, L.m);
    constexpr uint32_t RB = IS_SEQ ? 2048u : 16384u;
    constexpr uint32_t REGION = IS_SEQ ? 0u : SO_SBYTES;
    const uint32_t mode = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.mode);
    const uint32_t tus = mode & 0xFFu;
    const uint32_t nst = WIDE ? mode >> 16 : NSL;   // (wave-uniform) steps behind the last line's end are not issued
    const bool ragged_tails = tus == 0xFFu;
    uint32_t chk = 0, orw = 0, pt = 0;
    if (ragged_tails) {  // lines of different lengths in one batch: each lane picks the step of its line's partial dword
        uint32_t x = B.w[0];
#pragma unroll
        for (uint32_t u = 1; u < NSL; ++u) {
            x = S.tu == u ? B.w[u] : x;
            if (WIDE) asm volatile("" : "+v"(x));   // (a select per step: taken together the compiler makes them ONE indexed load — of a
                                                    // batch it then keeps in scratch memory, 80 bytes per lane, stored to at every update)
        }
        if (IS_SEQ) {
            pt = x & 0x07070707u;
            chk |= (x ^ __builtin_amdgcn_perm(0x474EFF54u, 0x43FF41FFu, pt)) & S.tb;
            orw |= x & S.tb;
        } else {
            pt = x - 0x21212121u;
            chk |= pt & S.tb;
        }
    }
    auto check_step = [&](auto U) {
        constexpr uint32_t u = decltype(U)::value;
        const uint32_t w = B.w[u], f = S.cm[u];
        if (IS_SEQ) {
            const uint32_t bins = w & 0x07070707u;
            chk |= (w ^ __builtin_amdgcn_perm(0x474EFF54u, 0x43FF41FFu, bins)) & f;
            orw |= w & f;
        } else {
            const uint32_t t = w - 0x21212121u;  // byte - 33 < 64 for all four bytes <=> bits 6-7 clear (stats_dev.h)
            chk |= t & f;
            B.w[u] = t;
        }
    };
    fz_steps<NSL, WIDE>(nst, check_step);
    if (__ballot(IS_SEQ ? chk != 0 : (chk & 0xC0C0C0C0u) != 0) != 0) {
        fz_dump<IS_SEQ, NSL>(B, z);
        return;
    }
    auto count_step = [&](auto U) {
        constexpr uint32_t u = decltype(U)::value;
        const uint32_t pb = IS_SEQ ? B.w[u] & 0x07070707u : B.w[u];
        // PACK: an odd step counts in the upper half of its word (subtracting 0xFFFF0000 adds 0x10000)
        const uint32_t f = (PACK && (u & 1u)) ? S.cm[u] & 0xFFFF0000u : S.cm[u];
        // (the row block and slot half go into the instruction's immediate offset)
        constexpr uint32_t off_u = fz_off<WIDE>(REGION, RB, u);
        if (u == tus) {  // (wave-uniform) the step that also holds the partial last dwords: per-byte values.  ADDs of 0 / 1, so
            // that the compiler cannot merge the two arms into one with four v_mov / v_cndmask per step in front of it
#pragma unroll
            for (int k = 0; k < 4; ++k)
                (void)__hip_atomic_fetch_add((so_lds_u32 *)(uintptr_t)(__builtin_amdgcn_perm(c.slots, pb, c.sel[k]) + off_u),
                                             (PACK && (u & 1u)) ? S.tv[k] << 16 : S.tv[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            fz_sub4_at<IS_SEQ>(off_u, c, pb, f, f, f, f);
        }
    };
    fz_steps<NSL, WIDE>(nst, count_step);
    if (ragged_tails) {
        const uint32_t off = WIDE ? REGION + (S.tu >> 1) * RB : REGION + ((S.tu & 1u) << 7) + (S.tu >> 1) * RB;
        const uint32_t sh = PACK ? (S.tu & 1u) << 4 : 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            (void)__hip_atomic_fetch_add((so_lds_u32 *)(uintptr_t)(__builtin_amdgcn_perm(c.slots, pt, c.sel[k]) + off), S.tv[k] << sh,
                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (IS_SEQ) {  // sequence lines with an 'N' (bit 3 is set in 'N' only): the 8 lanes of a line OR their flags
        const unsigned long long bn = __ballot((orw & 0x08080808u) != 0);
        if (bn) T.not_dna += WIDE ? so_groups16(bn) : so_groups(bn);
    }
}

// The lines that ended in this chunk of entries join the batches in progress, one per kind (sequence, quality).  Pent:
// lane j holds the packed word of the line that entry j of the chunk closes; a kind's lines are at lanes l0, l0 + 4, ..
// (n of them).  Position q = nfill + i of a kind's running sequence of lines: batch q / 8, slot q % 8.  Full batches are
// counted; fewer than eight lines stay in the kind's batch for the next call; flush counts them as well.
// Order of the LDS traffic of one iteration: the reads of both kinds' lines and the lookups of the NEXT iteration's
// lines first, the atomics of both kinds behind them — LDS operations of a wave complete in order, and a read that is
// issued behind 24 atomics waits for all of them.
template <uint32_t NSL>
struct FzRaw {
    fz_u32x2 v[NSL];
};
// (nrd: the steps to read — all of them but in the wide instance, which reads up to the longest line it has met so far)
template <uint32_t NSL, bool WIDE>
__device__ __forceinline__ void fz_issue(const FzBatch<NSL> &B, FzRaw<NSL> &R, const FzLane &L, const uint8_t *lds8, uint32_t nrd) {
    const uint32_t la = L.wm4 + ((B.P >> 16) & ~3u);
    fz_steps<NSL, WIDE>(nrd, [&](auto U) {
        constexpr uint32_t u = decltype(U)::value;
        R.v[u] = *reinterpret_cast<const fz_u32x2 *>(lds8 + la + (WIDE ? 64u : 32u) * u);
    });
}
template <uint32_t NSL, bool WIDE>
__device__ __forceinline__ void fz_align(FzBatch<NSL> &B, const FzRaw<NSL> &R, uint32_t nrd) {
    const uint32_t sh = (B.P >> 16) & 3u;
    fz_steps<NSL, WIDE>(nrd, [&](auto U) {
        constexpr uint32_t u = decltype(U)::value;
        B.w[u] = __builtin_amdgcn_alignbyte(R.v[u].y, R.v[u].x, sh);
    });
}
struct FzKind {          // one kind's lines of the chunk
    uint32_t l0, n;      // lanes l0, l0 + 4, ..: n lines
};
template <uint32_t NSL, bool WIDE>
__device__ __forceinline__ void fz_lines2(FzBatch<NSL> &PBs, uint32_t &nfs, FzKind ks, FzBatch<NSL> &PBq, uint32_t &nfq, FzKind kq,
                                          uint32_t Pent, bool flush, const FzLane &L, const uint8_t *lds8, FzShape<NSL> &S,
                                          SoTotals &T, bool &bad, const FusedArgs &z, bool do_count, uint32_t &lmx) {
    constexpr uint32_t LPB = WIDE ? 4u : 8u, LPB_SH = WIDE ? 2u : 3u;   // lines per batch
    const uint32_t q0s = nfs, tots = q0s + ks.n, nbs = tots >> LPB_SH, rems = tots & (LPB - 1u), nits = nbs + ((rems && flush) ? 1u : 0u);
    const uint32_t q0q = nfq, totq = q0q + kq.n, nbq = totq >> LPB_SH, remq = totq & (LPB - 1u), nitq = nbq + ((remq && flush) ? 1u : 0u);
    const uint32_t nbm = nbs > nbq ? nbs : nbq;
    int is0 = -(int)q0s, iq0 = -(int)q0q;  // slot g8 of batch b takes new line i = 8 b - q0 + g8, held by lane l0 + 4 i
    uint32_t Pns = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(4u * ks.l0 + 16u * (uint32_t)is0 + L.g16), (int)Pent);
    uint32_t Pnq = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(4u * kq.l0 + 16u * (uint32_t)iq0 + L.g16), (int)Pent);
    for (uint32_t b = 0; b <= nbm; ++b, is0 += (int)LPB, iq0 += (int)LPB) {
        const bool act_s = b <= nbs, act_q = b <= nbq;  // (wave-uniform) the kind still assembles a batch in this iteration
        const bool news = act_s && (uint32_t)(is0 + (int)L.g8) < ks.n;  // (a negative index wraps: the unsigned compare rejects it)
        const bool newq = act_q && (uint32_t)(iq0 + (int)L.g8) < kq.n;
        if (act_s && (b != 0 || L.g8 >= q0s)) PBs.P = news ? Pns : 0u;  // (slots below q0 of the first batch keep their lines)
        if (act_q && (b != 0 || L.g8 >= q0q)) PBq.P = newq ? Pnq : 0u;
        FzRaw<NSL> Rs, Rq;
        uint32_t nrd = NSL;
        if (WIDE && FQH_WIDE_NRD) {   // lmx (wave-uniform): the longest line this wavefront has met; steps behind it are neither read nor aligned
            const uint32_t ls = news ? PBs.P & 0x1FFu : 0u, lq = newq ? PBq.P & 0x1FFu : 0u;
            uint32_t ln = ls > lq ? ls : lq;
            if (__ballot(ln > lmx) != 0) {
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) {
                    const uint32_t o = (uint32_t)__shfl_xor((int)ln, d);
                    ln = o > ln ? o : ln;
                }
                lmx = (uint32_t)__builtin_amdgcn_readfirstlane((int)ln);
            }
            nrd = (lmx + 63u) >> 6;
        }
        if (news) fz_issue<NSL, WIDE>(PBs, Rs, L, lds8, nrd);
        if (newq) fz_issue<NSL, WIDE>(PBq, Rq, L, lds8, nrd);
        Pns = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(4u * ks.l0 + 16u * (uint32_t)(is0 + (int)LPB) + L.g16), (int)Pent);
        Pnq = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(4u * kq.l0 + 16u * (uint32_t)(iq0 + (int)LPB) + L.g16), (int)Pent);
        if (news) fz_align<NSL, WIDE>(PBs, Rs, nrd);
        if (act_s && b < nits && do_count) fz_count<true, NSL, WIDE>(PBs, S, L, T, bad, z);
        if (newq) fz_align<NSL, WIDE>(PBq, Rq, nrd);
        if (act_q && b < nitq && do_count) fz_count<false, NSL, WIDE>(PBq, S, L, T, bad, z);
    }
    nfs = flush ? 0u : rems;
    nfq = flush ? 0u : remq;
    if (flush) PBs.P = PBq.P = 0;
}

// 16 bytes at buf + off of the partial tile at the end of the buffer; bytes at or beyond len read as 0
__device__ __attribute__((noinline)) uint4 fz_load16_tail(const uint8_t *__restrict__ buf, uint64_t off, uint64_t len) {
    return load16(buf, off, len);
}

template <uint32_t NSL, uint32_t FZ_WAVES, bool WIDE>
__global__ __launch_bounds__(FZ_WAVES * 64) void k_scan_stats(FusedArgs z) {
    constexpr bool PACK = WIDE;
    constexpr uint32_t FZ_THREADS = FZ_WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];
    const uint32_t lc = z.lc;
    const uint32_t wb0 = z.wave_base;  // bytes of histogram in front of the waves' areas
    for (uint32_t i = threadIdx.x; i < wb0 / 4; i += FZ_THREADS) hist[i] = 0;
    __syncthreads();
    // The address registers assume the histogram starts at LDS address 0 (the kernel's only LDS object).
    if ((uint32_t)(uintptr_t)hist != 0) __builtin_trap();
    uint8_t *const lds8 = reinterpret_cast<uint8_t *>(hist);

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t wbase = wb0 + wv * FZ_WAVE_BYTES;  // LDS address of y = 0 of the wave's data area
    FzLane L;
    L.m = WIDE ? lane & 15u : lane & 7u;
    L.wm4 = wbase + L.m * 4u;
    L.g8 = WIDE ? lane >> 4 : lane >> 3;
    L.g16 = L.g8 * 16u;
    L.c.slots = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t j = k ^ (WIDE ? L.g8 & 1u : L.g8 & 3u);
        L.c.sel[k] = 0x0C0C0004u + k + (j << 8);
        L.c.slots |= ((WIDE ? L.m + 16u * j : L.m + 8u * j) * 4u) << (8u * k);
    }
    uint8_t *const wptr = lds8 + wbase + FZ_TAIL + 16u * lane;      // chunk 64 j + lane of the group: + 1024 j
    const uint32_t rbase = wbase + FZ_TAIL + 64u * lane;            // this lane's 64 contiguous bytes
    const uint8_t *const rptr = lds8 + rbase;
    // conflict-free read-back of the linear image: instruction i reads chunk (i + rot) % 4 of the lane
    const uint32_t rot = (lane >> 2) & 3u;
    const uint32_t kro = (4u - rot) & 3u;                           // rotate the 64-bit mask right by 16 kro bits
    const bool swp = (kro & 2u) != 0;
    const uint32_t s16 = (kro & 1u) * 16u;
    uint16_t *const tedge = reinterpret_cast<uint16_t *>(lds8 + wbase + FZ_DATA);  // the tile's first four entries
    uint16_t *const trs = tedge + 8;       // the tile's record starts (FZ_RS of them)
    uint16_t *const lst = trs + FZ_RS + 4; // the group's entries; lst[-4 .. -1]: the last four before the group

    uint32_t acc_rec = 0, acc_bases = 0, acc_qual = 0;  // per lane (a wave never reads 4 GiB)
    SoTotals T = {0, 0};
    FzShape<NSL> S = {};
    S.key = 0xFFFFFFFFu;

    const uint8_t *__restrict__ const buf = z.buf;
    const uint64_t len = z.len;
    const uint32_t n_tiles = (uint32_t)z.n_tiles;
    const uint32_t n_full = (uint32_t)(len >> WT_SHIFT);
    // (a partial tile of at most one group at the end of the buffer rides with the span in front of it: too few line
    // starts to settle an alignment of its own)
    const uint32_t n_spans = fz_spans(n_tiles, len);
    const uint32_t nw = gridDim.x * FZ_WAVES;
    const uint32_t lo = lane * 16u;
    uint32_t n_over = 0;
#ifdef FQH_FZ_TIMING
    unsigned long long tph[6] = {0, 0, 0, 0, 0, 0}, tk = 0;
#define FZ_T(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); tph[i] += t_ - tk; tk = t_; } while (0)
#else
#define FZ_T(i) do { } while (0)
#endif

    // the four 16-byte pieces of this lane for group g of tile t (whole tiles: unconditional loads)
    auto fetch_group = [&](uint32_t t, uint32_t g, uint4 &n0, uint4 &n1, uint4 &n2, uint4 &n3) {
        const uint64_t off = ((uint64_t)t << WT_SHIFT) + g * FZ_GROUP + lo;
        if (t < n_full) {
            const uint8_t *p = buf + off;
            n0 = load16_nt(p); n1 = load16_nt(p + PIECE_BYTES);
            n2 = load16_nt(p + 2 * PIECE_BYTES); n3 = load16_nt(p + 3 * PIECE_BYTES);
        } else {  // the partial tile at the end of the buffer
            n0 = fz_load16_tail(buf, off, len); n1 = fz_load16_tail(buf, off + PIECE_BYTES, len);
            n2 = fz_load16_tail(buf, off + 2 * PIECE_BYTES, len); n3 = fz_load16_tail(buf, off + 3 * PIECE_BYTES, len);
        }
    };

    // PACK: the block's 16-bit halves -> its 32-bit rows in scratch (zeroed by the host), the LDS cleared; every wavefront of the
    // block calls it the same number of times
    auto flush_rows = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        uint32_t *__restrict__ rows32 = z.scratch + (uint64_t)blockIdx.x * (2u * SO_WORDS);
        for (uint32_t i = threadIdx.x; i < wb0 / 4; i += FZ_THREADS) {
            const uint32_t v = hist[i];
            if (v) {
                if (v & 0xFFFFu) rows32[2u * i] += v & 0xFFFFu;
                if (v >> 16) rows32[2u * i + 1u] += v >> 16;
                hist[i] = 0;
            }
        }
        __syncthreads();
    };
    uint32_t lmx = 0;        // WIDE: the longest sequence / quality line this wavefront has met (columns)
    constexpr uint32_t EPOCH = fz_epoch(FZ_WAVES);
    uint32_t it = 0;         // PACK: spans this wavefront has begun (the block flushes when it is a multiple of EPOCH)

    uint32_t span = blockIdx.x * FZ_WAVES + wv;
    if (span < n_spans) {
        uint4 n0, n1, n2, n3;
        fetch_group(span * FZ_SPAN, 0, n0, n1, n2, n3);
        // the byte before the span, kept in a vector register until the span starts (no wait right behind the load)
        uint32_t pbv = span ? buf[((uint64_t)(span * FZ_SPAN) << WT_SHIFT) - 1] : 0u;
        bool pending = false;  // the previous tile's line is still in a register
        uint32_t ptile = 0, prv = 0;
        for (; span < n_spans; span += nw) {
            if (PACK) {
                if (it && it % EPOCH == 0) flush_rows();
                ++it;
            }
            const uint32_t nspan = span + nw < n_spans ? span + nw : span;  // clamped: the prefetch is unconditional
            const uint32_t t0 = span * FZ_SPAN;
            const uint32_t t1 = span + 1 == n_spans ? n_tiles : t0 + FZ_SPAN;
            uint32_t srun = 0;       // entries of the span before the current group
            uint32_t tot = 0;        // entries of the current group
            uint32_t prev;           // the byte before the group is a newline
            {
                uint32_t x = pbv;
                asm volatile("v_mov_b32 %0, %0" : "+v"(x));  // (opaque: keeps the compiler from moving the value to a scalar at the load)
                prev = (t0 && (uint32_t)__builtin_amdgcn_readfirstlane((int)x) == '\n') ? 1u : 0u;
            }
            uint32_t hyp = 7;        // the span's alignment: entries hyp, hyp + 4, .. (counted from the span's first) start records
            bool span_bad = false;
            FzBatch<NSL> PBs, PBq;   // the batches in progress
            PBs.P = 0;
            PBq.P = 0;
#pragma unroll
            for (uint32_t u = 0; u < NSL; ++u) PBs.w[u] = PBq.w[u] = 0;
            uint32_t nfill_s = 0, nfill_q = 0;
            uint2 post = make_uint2(0, 0);
            for (uint32_t tile = t0; tile < t1; ++tile) {
                const uint64_t tb = (uint64_t)tile << WT_SHIFT;
                const bool full = tile < n_full;
                const uint32_t tile_bytes = full ? WT_BYTES : (uint32_t)(len - tb);
                const uint32_t ng = full ? WT_BYTES / FZ_GROUP : (tile_bytes + FZ_GROUP - 1) / FZ_GROUP;
                const bool last_t = tile + 1 == t1;
                uint32_t run = 0;            // entries of the tile before the current group
                uint32_t have = 0, bad = 0;  // bit r: some / some failing window of five entries starting at r (mod 4), tile-relative
                uint32_t hyp_t = hyp < 4 ? (hyp - srun - tot) & 3u : 7u;  // the same alignment counted from the tile's first entry
                __builtin_amdgcn_wave_barrier();
                trs[lane] = 0;  // (unused slots of the tile's line are 0)
#pragma unroll 1
                for (uint32_t g = 0; g < ng; ++g) {
                    const bool last_g = g + 1 == ng;
#ifdef FQH_FZ_TIMING
                    tk = __builtin_readcyclecounter();
#endif
                    __builtin_amdgcn_wave_barrier();
                    *reinterpret_cast<uint4 *>(wptr) = n0;
                    *reinterpret_cast<uint4 *>(wptr + 1024) = n1;
                    *reinterpret_cast<uint4 *>(wptr + 2048) = n2;
                    *reinterpret_cast<uint4 *>(wptr + 3072) = n3;
                    int ppos = -1;  // position of the first newline in the bytes after the span
                    if (last_g && last_t && full) {  // those bytes (loaded three groups ago) go behind the group
                        *reinterpret_cast<uint2 *>(lds8 + wbase + FZ_TAIL + FZ_GROUP + 8u * lane) = post;
                        const uint32_t f0 = eq_flags(post.x, 0x0A0A0A0Au), f1 = eq_flags(post.y, 0x0A0A0A0Au);
                        const unsigned long long bm = __ballot((f0 | f1) != 0);
                        if (bm) {
                            const uint32_t first = (uint32_t)__ffsll((long long)bm) - 1u;
                            const uint32_t g0 = (uint32_t)__builtin_amdgcn_readlane((int)f0, (int)first);
                            const uint32_t g1 = (uint32_t)__builtin_amdgcn_readlane((int)f1, (int)first);
                            const uint32_t byte = g0 ? ((uint32_t)__ffs(g0) - 1u) / 8u : 4u + ((uint32_t)__ffs(g1) - 1u) / 8u;
                            ppos = (int)(8u * first + byte);
                        }
                    }
                    // the next group: of this tile, of the span's next tile, or of the wave's next span
                    if (last_t && g == 0) {  // (early: the bytes after the span are needed in the span's last group)
                        const uint64_t pe = tb + tile_bytes + 8u * lane;
                        if (pe + 8 <= len) post = *reinterpret_cast<const uint2 *>(buf + pe);
                    }
                    if (last_g && last_t) pbv = buf[((uint64_t)(nspan * FZ_SPAN) << WT_SHIFT) - (nspan ? 1 : 0)];
                    fetch_group(last_g ? (last_t ? nspan * FZ_SPAN : tile + 1) : tile, last_g ? 0u : g + 1, n0, n1, n2, n3);
                    if (g == 0 && pending)  // a whole group before the next wait on vmcnt
                        __builtin_nontemporal_store((uint16_t)prv, FZ_KARG(fast_rs) + (uint64_t)ptile * FR_STRIDE + lane);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    FZ_T(0);  // LDS write, prefetch issue
                    // (this group's last 512 bytes, for the next group's tail: read now, written at the group's end)
                    const uint2 tailv = *reinterpret_cast<const uint2 *>(lds8 + wbase + FZ_GROUP + 8u * lane);
                    uint32_t m_lo, m_hi;
                    {
                        const uint4 d0 = *reinterpret_cast<const uint4 *>(rptr + ((rot * 16u) & 48u));
                        const uint4 d1 = *reinterpret_cast<const uint4 *>(rptr + ((rot * 16u + 16u) & 48u));
                        const uint4 d2 = *reinterpret_cast<const uint4 *>(rptr + ((rot * 16u + 32u) & 48u));
                        const uint4 d3 = *reinterpret_cast<const uint4 *>(rptr + ((rot * 16u + 48u) & 48u));
                        const uint32_t r_lo = eqmask16<1>(d0, 0x0A0A0A0Au) | (eqmask16<1>(d1, 0x0A0A0A0Au) << 16);
                        const uint32_t r_hi = eqmask16<1>(d2, 0x0A0A0A0Au) | (eqmask16<1>(d3, 0x0A0A0A0Au) << 16);
                        const uint32_t a_lo = swp ? r_hi : r_lo, a_hi = swp ? r_lo : r_hi;
                        m_lo = __builtin_amdgcn_alignbit(a_hi, a_lo, s16);
                        m_hi = __builtin_amdgcn_alignbit(a_lo, a_hi, s16);
                    }
                    // line starts: the byte after a newline
                    uint32_t ls_lo = (m_lo << 1) | wave_shr1(m_hi >> 31, prev);
                    uint32_t ls_hi = __builtin_amdgcn_alignbit(m_hi, m_lo, 31);
                    prev = ((uint32_t)__builtin_amdgcn_readlane((int)m_hi, 63)) >> 31;
                    if (!full) {  // a line start must be an existing byte
                        const int nv = (int)tile_bytes - (int)(g * FZ_GROUP + lane * 64u);
                        const uint32_t nvalid = nv < 0 ? 0u : nv > 64 ? 64u : (uint32_t)nv;
                        const unsigned long long keep = nvalid >= 64 ? ~0ull : ((1ull << nvalid) - 1ull);
                        ls_lo &= (uint32_t)keep;
                        ls_hi &= (uint32_t)(keep >> 32);
                    }
                    const uint32_t cl = __popc(ls_lo) + __popc(ls_hi);
                    const unsigned long long b1 = __ballot(cl >= 1), b2 = __ballot(cl >= 2), b3 = __ballot(cl >= 3);
                    uint32_t pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, 0));
                    pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b2, pre));
                    pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(b3 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b3, pre));
                    uint32_t gtot = (uint32_t)__popcll(b1) + (uint32_t)__popcll(b2) + (uint32_t)__popcll(b3);
                    if (__ballot(cl >= 4)) {
                        for (uint32_t k = 4;; ++k) {
                            const unsigned long long b = __ballot(cl >= k);
                            if (!b) break;
                            pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, pre));
                            gtot += (uint32_t)__popcll(b);
                        }
                    }
                    FZ_T(1);  // read-back, newline masks, prefix
                    run += tot;   // the previous group's entries are behind us now
                    srun += tot;
                    if (g == 0) run = 0;  // (they belonged to the previous tile)
                    if (gtot > FZ_GLIST) span_bad = true;  // lines shorter than ~22 bytes on average: left to the exact path
                    tot = gtot < FZ_GLIST ? gtot : FZ_GLIST;
                    const uint32_t ebase = g * FZ_GROUP + lane * 64u;
                    if (pre + cl <= FZ_GLIST) {  // (a lane whose entries would leave the list writes none: the span is bad anyway)
                        // offsets only; the '@' / '+' bits are added by the per-entry pass below (one LDS round trip there
                        // instead of one per iteration here)
                        uint16_t *dst = lst + pre;
                        while (ls_lo) {
                            const uint32_t q = __ffs(ls_lo) - 1;
                            ls_lo &= ls_lo - 1;
                            *dst++ = (uint16_t)(ebase + q);
                        }
                        while (ls_hi) {
                            const uint32_t q = __ffs(ls_hi) + 31;
                            ls_hi &= ls_hi - 1;
                            *dst++ = (uint16_t)(ebase + q);
                        }
                    }
                    FZ_T(2);  // staging
                    const int gofs = (int)FZ_TAIL - (int)(g * FZ_GROUP);  // tile offset -> y (position in the wave's data area)
                    // ---- the span's last group: its last line ends in another wavefront's span (or with the buffer); the
                    // 512 bytes after the span close it, as one more (virtual) entry behind the group's
                    uint32_t totv = tot;
                    if (last_g && last_t && !FZ_DBG(16u)) {
                        const int yend = (int)FZ_TAIL + (int)(tile_bytes - g * FZ_GROUP);  // y of the first byte after the span
                        const bool last_nl = full ? prev != 0 : buf[len - 1] == '\n';
                        int yclose = -1;
                        if (last_nl) {
                            yclose = yend;
                        } else if (full) {
                            if (ppos >= 0) yclose = yend + ppos + 1;
                            else if (tb + tile_bytes + FZ_POST <= len) {  // a line that goes on for more than 512 bytes after the span:
                                if (lane == 0) atomicAdd(&FZ_KARG(out)->stats_declined, 1ull);  // not counted here (no doubt about the parse)
                            }
                            // (else: no '\n' before the end of the buffer: not a line the parser delivers)
                        }
                        if (yclose >= 0) {
                            if (lane == 0) lst[tot] = (uint16_t)((uint32_t)(yclose - gofs) & 0x3FFFu);
                            totv = tot + 1;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    // ---- one lane per entry, 64 entries at a time: the window of five entries that ends at it (src/records.rs:141,
                    // 155,233 under each alignment), and the line it closes (the one entry p - 1 starts)
                    // (the span's last group runs the pass at least once, entries or not: the batches in progress are flushed there — a
                    // chunk that is not the file's last may end in a group without a single line start)
                    const uint32_t nent = totv ? totv : (last_g && last_t ? 1u : 0u);
#pragma unroll 1
                    for (uint32_t c0 = 0; c0 < nent; c0 += 64) {
                        const uint32_t p = c0 + lane;
                        const uint32_t ti = run + p;
                        uint32_t Pent = 0;      // the line entry p closes
                        bool toolong = false;
                        uint32_t e4 = 0, bcr = 0;
                        uint32_t l = 0;
                        int yc = 0;
                        // phase A: the entry, the first byte of the line it starts ('@' / '+': read_header / read_sep look at
                        // nothing else, src/records.rs:141,155) and the byte in front of the '\n' that closes the line before it
                        if (p < totv) {
                            e4 = lst[p];
                            yc = (int)(e4 & 0x3FFFu) + gofs;
                            if (p >= tot && yc < (int)FZ_TAIL) yc += (int)WT_BYTES;  // (the virtual entry's offset may have wrapped)
                            const uint8_t *at = lds8 + wbase + (uint32_t)yc;
                            const uint32_t b0 = at[0];
                            bcr = *(at - 2);
                            e4 |= (b0 == '@' ? 0x4000u : 0u) | (b0 == '+' ? 0x8000u : 0u);
                            if (p < tot) lst[p] = (uint16_t)e4;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        // phase B: the window of five entries that ends here, and the line entry p closes
                        if (p < totv) {
                            const uint32_t e3 = lst[(int)p - 1];
                            if (p < tot && !FZ_DBG(8u)) {
                                if (ti < 4) {
                                    tedge[ti] = (uint16_t)e4;  // the tile's first four entries
                                } else {
                                    const uint32_t e0 = lst[(int)p - 4], e1 = lst[(int)p - 3], e2 = lst[(int)p - 2];
                                    const bool ok = (e0 & 0x4000u) && (e2 & 0x8000u) &&
                                                    ((e2 & 0x3FFFu) - (e1 & 0x3FFFu)) == ((e4 & 0x3FFFu) - (e3 & 0x3FFFu));
                                    have |= 1u << (ti & 3u);
                                    bad |= ok ? 0u : 1u << (ti & 3u);
                                }
                                if (tot >= 4 && p + 4 >= tot) lst[(int)p - (int)tot] = (uint16_t)e4;  // the last four go in front of the next group's
                            }
                            if ((srun | p) != 0) {  // (the span's first entry closes a line that is not this wave's)
                                l = ((e4 - e3) & 0x3FFFu) - 1u;  // raw line, without its '\n'
                                const int ys = yc - 1 - (int)l;
                                if (ys < 0) {
                                    if (!fz_given_up(z)) atomicAdd(&FZ_KARG(out)->stats_declined, 1ull);  // began before the kept tail (longer than ~500 bytes): not counted here
                                } else {
                                    if (l && bcr == '\r') --l;  // trim_winline, src/records.rs:66-73
                                    // (longer than the histogram's rows: the span is bad if this turns out to be a sequence or a
                                    // quality line, below; header and separator lines are never read and may be longer)
                                    toolong = l > lc;
                                    Pent = (toolong ? 0u : l) | FZ_P_ACT | ((uint32_t)ys << 16);
                                }
                            }
                        }
                        FZ_T(3);  // per-entry pass
                        span_bad = __ballot(span_bad) != 0;
                        const bool head_chunk = span == 0 && (srun | c0) == 0 && FZ_KARG(skip_head);  // (wave-uniform) the chunk's very first entries
                        if (tile == t0 && g == 0 && c0 == 0) {  // the span's first entries must single out the alignment it is counted under
                            uint32_t cons = 0;
#pragma unroll
                            for (uint32_t r = 0; r < 4; ++r)
                                if (__ballot((have >> r) & 1u) && !__ballot((bad >> r) & 1u)) cons |= 1u << r;
                            if (cons && !(cons & (cons - 1))) hyp = hyp_t = (uint32_t)__ffs(cons) - 1;
                            else span_bad = true;
                        }
                        {
                            // lines, counted from the span's first entry: hyp (mod 4) header, + 1 sequence, + 2 separator, + 3 quality;
                            // entry p closes line srun + p - 1.  (Runs whatever span_bad says: nothing of a bad span is used, and
                            // a region that is skipped conditionally costs a wait for the loads in flight, see DESIGN.md.)
                            const uint32_t kd = (srun + p - 1u - hyp) & 3u;
                            // a chunk that begins inside a record (carry-in): the lines up to the chunk's first record start belong to
                            // the record in progress, which is counted as a whole by k_stats_edge (it began in front of the chunk)
                            if (head_chunk && hyp < 4 && p <= hyp) Pent = 0;
                            {   // a sequence / quality line beyond the histogram's rows joins its batch with length 0 and is LISTED
                                // (where it starts in the buffer, its length, its kind) for k_stats_declined
                                const bool listed = toolong && (kd & 1u) && Pent != 0;
                                const unsigned long long lb = __ballot(listed);
                                if (lb && !fz_given_up(z)) {
                                    uint32_t base = 0;
                                    const uint32_t nl_ = (uint32_t)__popcll(lb);
                                    if (lane == 0) base = (uint32_t)atomicAdd(&FZ_KARG(out)->decl_lines, (unsigned long long)nl_);
                                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                                    if (base + nl_ > FZ_KARG(decl_cap)) {
                                        if (lane == 0) atomicAdd(&FZ_KARG(out)->stats_declined, 1ull);
                                    } else if (listed) {
                                        const uint32_t k = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(lb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lb, 0));
                                        uint32_t *e = reinterpret_cast<uint32_t *>(FZ_KARG(decl_l)) + 4u * k;   // (tile, y | group << 16, length, kind)
                                        e[0] = tile;
                                        e[1] = (Pent >> 16) | (g << 16);
                                        e[2] = l;
                                        e[3] = kd == 3u ? 1u : 0u;
                                    }
                                }
                            }
                            if (Pent) {
                                if (kd == 1u) { ++acc_rec; acc_bases += l; }
                                if (kd == 3u) acc_qual += l;
                            }
                            if (kd == 3u && p < tot && ti >= hyp_t) {  // a record starts here: record start k of the tile
                                const uint32_t k = (ti - hyp_t) >> 2;
                                if (k < FZ_RS) trs[k] = (uint16_t)(e4 & 0x3FFFu);
                            }
                            const uint32_t cnt_c = totv - c0 < 64 ? totv - c0 : 64u;
                            const uint32_t pq = (hyp - srun - c0) & 3u;  // lanes pq, pq + 4, ..: entries that close a quality line
                            uint32_t ps = (pq + 2u) & 3u;                // lanes ps, ps + 4, ..: entries that close a sequence line
                            uint32_t pq0 = pq;
                            if ((srun | c0) == 0) {  // (the span's very first entry closes nothing of ours)
                                if (ps == 0) ps = 4;
                                if (pq0 == 0) pq0 = 4;
                            }
                            if (head_chunk && hyp < 4) {  // (nor do the entries up to the chunk's first record start)
                                if (ps <= hyp) ps += 4;
                                if (pq0 <= hyp) pq0 += 4;
                            }
                            const uint32_t nls = cnt_c > ps ? (cnt_c - ps + 3) >> 2 : 0u, nlq = cnt_c > pq0 ? (cnt_c - pq0 + 3) >> 2 : 0u;
                            const bool flush = last_g && last_t && c0 + 64 >= totv;
                            const bool cnt = !FZ_DBG(2u);
                            FZ_T(5);
                            fz_lines2<NSL, WIDE>(PBs, nfill_s, FzKind{ps, nls}, PBq, nfill_q, FzKind{pq0, nlq}, Pent, flush, L, lds8, S, T, span_bad, z, cnt, lmx);
                            FZ_T(4);  // lines: lookups, reads, counts
                        }
                    }
                    // ---- the next group finds this one's last 512 bytes (and, above, its last four entries) in front of its own
                    __builtin_amdgcn_wave_barrier();
                    if (tot < 4) {  // (rare: the four entries in front of the next group are partly the old ones)
                        const uint32_t hv = lane < 4 ? (uint32_t)lst[(int)tot - 4 + (int)lane] : 0u;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        if (lane < 4) lst[(int)lane - 4] = (uint16_t)hv;
                    }
                    *reinterpret_cast<uint2 *>(lds8 + wbase + 8u * lane) = tailv;
                    FZ_T(5);  // the rest
                }
                const uint32_t trun = run + tot;  // entries of the whole tile
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // ---- the tile must confirm the alignment: still the only consistent one
                // (the short tile at the end of the buffer cannot: it is counted under the span's alignment, marked FR_SMALL,
                // and k_finalize_fast validates its records against the true line index)
                const bool small_t = !full && trun < 8;
                {
                    uint32_t cons = 0, badb = 0;
#pragma unroll
                    for (uint32_t r = 0; r < 4; ++r) {
                        const bool b = __ballot((bad >> r) & 1u) != 0;
                        badb |= b ? 1u << r : 0u;
                        if (__ballot((have >> r) & 1u) && !b) cons |= 1u << r;
                    }
                    if (small_t) {
                        if (hyp_t > 3 || ((badb >> hyp_t) & 1u)) span_bad = true;
                    } else if (trun < 8 || hyp_t > 3 || cons != (1u << hyp_t)) {
                        span_bad = true;
                    }
                }
                // ---- the tile's line: record starts (the lanes staged them), first and last four entries, count, alignment;
                // record starts beyond the line's FR_N go to a second line and to the list area (reads shorter than ~140 bp)
                {
                    uint32_t rv = lane < FR_N ? (uint32_t)trs[lane] : 0u;
                    if (lane >= FR_EDGE && lane < FR_EDGE + 4) rv = tedge[lane - FR_EDGE];
                    if (lane >= FR_EDGE + 4 && lane < FR_EDGE + 8) rv = lst[(int)lane - (int)(FR_EDGE + 8)];
                    if (small_t) {  // entries 0 .. trun - 1 in order, no record starts
                        const uint32_t k = lane - FR_EDGE;
                        rv = (lane >= FR_EDGE && k < trun) ? (k < 4 ? (uint32_t)tedge[k] : (uint32_t)lst[(int)k - (int)trun]) : 0u;
                    }
                    const uint32_t nrs = trun > hyp_t ? (trun - hyp_t + 3) >> 2 : 0u;  // record starts of the tile
                    if (nrs > FR_N && hyp_t < 4) {
                        FZ_KARG(fast_rs)[fr2_off(n_tiles) + (uint64_t)tile * FR2_N + lane] = trs[FR_N + lane];
                        if (nrs > FR_N + FR2_N) {
                            if (FZ_KARG(list)) {
                                for (uint32_t k = FR_N + FR2_N + lane; k < nrs && k < FZ_RS; k += 64) FZ_KARG(list)[(uint64_t)tile * FZ_KARG(list_cap) + 8 + k] = trs[k];
                            } else {  // (no line-list workspace yet: the host reruns with it)
                                span_bad = true;
                                if (lane == 0) FZ_KARG(out)->need_list = 1;
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
                        trs[64 + lane] = 0;
                        trs[128 + lane] = 0;
                        trs[192 + lane] = 0;
                    }
                    if (span_bad) ++n_over;
                    prv = lane == FR_CNT ? (trun & 0xFFFFu) : lane == FR_CNT + 1 ? (trun >> 16) : lane == FR_HYP ? (span_bad ? 7u : small_t ? (FR_SMALL | hyp_t) : hyp_t) : rv;
                }
                ptile = tile;
                pending = true;
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (pending) __builtin_nontemporal_store((uint16_t)prv, FZ_KARG(fast_rs) + (uint64_t)ptile * FR_STRIDE + lane);
    }
    if (lane == 0 && n_over) atomicAdd(&FZ_KARG(out)->spec_fail, (unsigned long long)n_over);
#ifdef FQH_FZ_TIMING
    if (lane == 0)
        for (int i = 0; i < 6; ++i) atomicAdd(&FZ_KARG(scalars)[8 + i], tph[i]);
#endif

    // ---- per-block partial histogram, per-wave totals
    if (PACK) {   // (the wavefronts that ran out of spans early take part in the flushes of the others, then in the last one)
        const uint32_t rounds = (n_spans + nw - 1) / nw;
        for (; it < rounds; ++it)
            if (it && it % EPOCH == 0) flush_rows();
        flush_rows();
    } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        uint32_t *__restrict__ dst = FZ_KARG(scratch) + (uint64_t)blockIdx.x * SO_WORDS;
        for (uint32_t i = threadIdx.x; i < wb0 / 4; i += FZ_THREADS) dst[i] = hist[i];
    }
    unsigned long long sc[5] = {acc_rec, acc_bases, acc_qual, 0, 0};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        unsigned long long v = sc[j];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
        sc[j] = v;
    }
    if (lane == 0) {
        sc[3] = sc[0] - T.not_dna;
        sc[4] = sc[0] - T.not_dnan;
#pragma unroll
        for (int j = 0; j < 5; ++j)
            if (sc[j]) atomicAdd(&FZ_KARG(scalars)[j], sc[j]);
    }
}

// k_stats_commit: adds what k_scan_stats left in scratch — per-block partial histograms (bank-scheduled layout)
// and the totals — to the caller's arrays if and only if the scan's finalize kernel kept the fast path's result
// (DevOut::stats_commit).
// (lc: the rows the pass kept; lmax: the rows the caller has.  The pass may keep MORE than the caller has — reads longer than lmax:
// what it counted in rows lmax .. lc - 1 are the columns beyond lmax, scalars[5] / [6])
__device__ __forceinline__ void commit_overflow(unsigned long long over_s, unsigned long long over_q, unsigned long long *__restrict__ scalars) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        over_s += __shfl_xor(over_s, d);
        over_q += __shfl_xor(over_q, d);
    }
    if ((threadIdx.x & 63u) == 0) {
        if (over_s) atomicAdd(&scalars[5], over_s);
        if (over_q) atomicAdd(&scalars[6], over_q);
    }
}
__global__ __launch_bounds__(256) void k_stats_commit(const DevOut *__restrict__ out, const uint32_t *__restrict__ scratch,
                                                      uint32_t n_blocks, uint32_t lc, uint32_t lmax, uint32_t words,
                                                      const unsigned long long *__restrict__ src_scalars,
                                                      unsigned long long *__restrict__ qual_hist,
                                                      unsigned long long *__restrict__ base_hist,
                                                      unsigned long long *__restrict__ scalars) {
    if (!out->stats_commit) return;
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.y == 0 && id < FQH_NSCALARS && src_scalars[id]) atomicAdd(&scalars[id], src_scalars[id]);
    const bool isq = id >= SO_SBYTES / 4;
    const uint32_t r = isq ? id - SO_SBYTES / 4 : id;
    const uint32_t rb = isq ? r >> 12 : r >> 9;
    const uint32_t bin = isq ? (r >> 6) & 63u : (r >> 6) & 7u;
    const uint32_t row = rb * 64 + so_row6(r & 63u);
    unsigned long long s = 0;
    if (id < words && row < lc) {
        const uint32_t b0 = blockIdx.y * RED_GROUP;
        const uint32_t b1 = b0 + RED_GROUP < n_blocks ? b0 + RED_GROUP : n_blocks;
        for (uint32_t b = b0; b < b1; ++b) s += scratch[(uint64_t)b * SO_WORDS + id];
    }
    if (s && row < lmax) {
        if (isq) atomicAdd(&qual_hist[(uint64_t)row * 256 + 33 + bin], s);
        else atomicAdd(&base_hist[(uint64_t)row * 8 + bin_to_class(bin)], s);  // bins 0,2,5 share class 5
        s = 0;
    }
    if (lc > lmax) commit_overflow(isq ? 0ull : s, isq ? s : 0ull, scalars);   // (launch-uniform; every lane is here)
}

// ... and the wide instance's rows: the blocks' 32-bit rows in scratch, [2 i + h] = half h of LDS word i (k_scan_stats<8, 12, true>:
// step u of a line — 64 columns — counts in half u & 1 of row block u >> 1, slot m + 16 j for row 64 u + 4 m + j)
__global__ __launch_bounds__(256) void k_stats_commit_packed(const DevOut *__restrict__ out, const uint32_t *__restrict__ scratch,
                                                             uint32_t n_blocks, uint32_t lc, uint32_t lmax, const unsigned long long *__restrict__ src_scalars,
                                                             unsigned long long *__restrict__ qual_hist,
                                                             unsigned long long *__restrict__ base_hist,
                                                             unsigned long long *__restrict__ scalars) {
    if (!out->stats_commit) return;
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.y == 0 && id < FQH_NSCALARS && src_scalars[id]) atomicAdd(&scalars[id], src_scalars[id]);
    const uint32_t w = id >> 1, half = id & 1u;
    const bool isq = w >= SO_SBYTES / 4;
    const uint32_t r = isq ? w - SO_SBYTES / 4 : w;
    const uint32_t rb = isq ? r >> 12 : r >> 9;
    const uint32_t bin = isq ? (r >> 6) & 63u : (r >> 6) & 7u;
    const uint32_t slot = r & 63u;
    const uint32_t row = rb * 128u + half * 64u + (slot & 15u) * 4u + (slot >> 4);
    unsigned long long s = 0;
    if (id < 2u * SO_WORDS && row < lc) {
        const uint32_t b0 = blockIdx.y * RED_GROUP;
        const uint32_t b1 = b0 + RED_GROUP < n_blocks ? b0 + RED_GROUP : n_blocks;
        for (uint32_t b = b0; b < b1; ++b) s += scratch[(uint64_t)b * (2u * SO_WORDS) + id];
    }
    if (s && row < lmax) {
        if (isq) atomicAdd(&qual_hist[(uint64_t)row * 256 + 33 + bin], s);
        else atomicAdd(&base_hist[(uint64_t)row * 8 + bin_to_class(bin)], s);  // bins 0,2,5 share class 5
        s = 0;
    }
    if (lc > lmax) commit_overflow(isq ? 0ull : s, isq ? s : 0ull, scalars);   // (launch-uniform; every lane is here)
}

// k_stats_edge — the two records at the edges of a CHUNK that the single pass cannot count by itself (one wavefront, runs
// only if k_finalize_fast kept the fast path's result, like k_stats_commit):
//   * sign +1, the record in progress at the chunk start: it began `back0` bytes in front of the chunk (the caller's buffer
//     holds them: fqh_stats_launch_lead, the streaming ring) and is delivered with the chunk it ends in (src/lib.rs:255-303:
//     the reference keeps the partial record at the front of its Buffer and parses it again with the next read).
//     k_scan_stats skipped its lines (FusedArgs::skip_head); here all of it is counted, if it ends inside the chunk;
//   * sign -1, the partial record behind the last complete one of a chunk that is not the file's last: k_scan_stats counts
//     every sequence line that closes inside the buffer, also the one of a record whose quality line has not arrived yet —
//     that record is the NEXT chunk's (as its record in progress), so its sequence line is taken out again.
// The record is walked from its first byte: newlines by ballot over 64 bytes at a time, then the plain per-byte statement
// on the caller's u64 arrays.
__global__ __launch_bounds__(64) void k_stats_edge(const DevOut *__restrict__ out, const uint8_t *__restrict__ buf, uint64_t len,
                                                   uint64_t back0, int sign, uint32_t lmax,
                                                   unsigned long long *__restrict__ qual_hist,
                                                   unsigned long long *__restrict__ base_hist,
                                                   unsigned long long *__restrict__ scalars) {
    if (!out->stats_commit) return;
    const uint32_t lane = threadIdx.x;
    long long start, end = (long long)len;
    if (sign > 0) {
        if (out->n_records == 0) return;       // the record in progress does not end in this chunk: the next one counts it
        start = -(long long)back0;
    } else {
        start = out->end_off;                   // behind the last complete record
        if (start < 0 || start >= end) return;
    }
    long long nl[4];
    int found = 0;
    for (long long b = start; b < end && found < 4; b += 64) {
        const long long i = b + lane;
        unsigned long long m = __ballot(i < end && buf[i] == '\n');
        while (m && found < 4) {
            nl[found++] = b + (long long)__ffsll((long long)m) - 1;
            m &= m - 1;
        }
    }
    const int need = sign > 0 ? 4 : 2;
    if (found < need) return;
    if (sign < 0 && found >= 4) return;         // (cannot be: the record would have been complete)
    const unsigned long long one = sign > 0 ? 1ull : ~0ull;
    unsigned long long n_bases = 0, n_qual = 0, oseq = 0, oqual = 0;
    uint32_t any_n = 0, any_inv = 0;
    for (int kind = 0; kind < (sign > 0 ? 2 : 1); ++kind) {
        const long long s = (kind ? nl[2] : nl[0]) + 1;
        long long l = (kind ? nl[3] : nl[1]) - s;                 // raw line, without its '\n'
        if (l > 0 && buf[s + l - 1] == '\r') --l;                 // trim_winline, src/records.rs:66-73
        if (kind) n_qual = (unsigned long long)l; else n_bases = (unsigned long long)l;
        for (long long col = lane; col < l; col += 64) {
            const uint32_t b = buf[s + col];
            if (kind == 0) {
                const uint32_t c = base_class(b);
                any_inv |= c == 5 ? 1u : 0u;
                any_n |= c == 4 ? 1u : 0u;
                if (col < (long long)lmax) atomicAdd(&base_hist[(uint64_t)col * 8 + c], one);
                else ++oseq;
            } else {
                if (col < (long long)lmax) atomicAdd(&qual_hist[(uint64_t)col * 256 + b], one);
                else ++oqual;
            }
        }
    }
    const bool gi = __ballot(any_inv != 0) != 0, gn = __ballot(any_n != 0) != 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        oseq += __shfl_xor(oseq, d);
        oqual += __shfl_xor(oqual, d);
    }
    if (lane == 0) {
        atomicAdd(&scalars[0], one);
        if (n_bases) atomicAdd(&scalars[1], one * n_bases);
        if (n_qual) atomicAdd(&scalars[2], one * n_qual);
        if (!gi && !gn) atomicAdd(&scalars[3], one);
        if (!gi) atomicAdd(&scalars[4], one);
        if (oseq) atomicAdd(&scalars[5], one * oseq);
        if (oqual) atomicAdd(&scalars[6], one * oqual);
    }
}
void launch_stats_edge(hipStream_t s, const DevOut *out, const uint8_t *buf, uint64_t len, uint64_t back0, int sign, uint32_t lmax,
                       unsigned long long *qual_hist, unsigned long long *base_hist, unsigned long long *scalars) {
    hipLaunchKernelGGL(k_stats_edge, dim3(1), dim3(64), 0, s, out, buf, len, back0, sign, lmax, qual_hist, base_hist, scalars);
}

// k_stats_declined — what k_scan_stats could not count itself (fz_dump, the list of long lines), counted with the plain
// per-byte statement; runs behind k_stats_commit and, like it, only if the pass is committed (DevOut::stats_commit).  One
// wavefront per dumped batch / listed line.  A dumped batch is eight whole lines — a thousand clean bytes around the one that
// made the kernel decline — so the counts first go to a histogram in the block's LDS (lmax rows x (8 classes + 256 quality
// values), 16-bit counters packed in pairs: a block never sees more than DECL_LINES_PER_BLOCK lines) and only its non-zero
// bins to the caller's u64 arrays: 18 M global atomics on the same few thousand addresses took 1.5 ms for one dirty byte in a
// million, the flush of 128 small histograms takes a twentieth of that.  The single pass has already counted these lines'
// records and lengths (its totals come from the line starts), and has taken every sequence line it did not count itself for
// valid DNA: a line with an 'N' or a byte outside ACGTN is taken out of scalars[3] / [4] here (validate_dna / validate_dnan,
// src/records.rs:19-33).
constexpr uint32_t DECL_BINS = 264;               // per row: base classes 0..7, then quality values 0..255
constexpr uint32_t DECL_LINES_PER_BLOCK = 60000;  // (16-bit counters)
__global__ __launch_bounds__(1024) void k_stats_declined(const DevOut *__restrict__ out, const uint32_t *__restrict__ decl_b, uint32_t nsl,
                                                        const uint64_t *__restrict__ decl_l, uint32_t cap, const uint8_t *__restrict__ buf,
                                                        uint32_t lmax, uint32_t lc, uint32_t row0, uint32_t rows, uint32_t wide,
                                                        unsigned long long *__restrict__ qual_hist,
                                                        unsigned long long *__restrict__ base_hist, unsigned long long *__restrict__ scalars) {
    // lc: the rows the caller's arrays and the single pass share; this launch counts columns row0 .. row0 + rows - 1 of them (a
    // window of at most 256 rows fits the LDS; the launch with row0 == 0 also settles the lines' alphabet verdicts and the
    // columns beyond lc); wide: the batches were dumped by the wide instance (sixteen lanes per line, 64 columns per step)
    if (!out->stats_commit) return;
    const uint32_t nb = (uint32_t)(out->decl_b < cap ? out->decl_b : cap), nl = (uint32_t)(out->decl_l < cap ? out->decl_l : cap);
    if (!nb && !nl) return;
    extern __shared__ __attribute__((aligned(16))) uint32_t dh[];   // rows * DECL_BINS / 2 words
    const uint32_t words = rows * DECL_BINS / 2;
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) dh[i] = 0;
    __syncthreads();
    auto count = [&](uint32_t row, uint32_t bin) {
        const uint32_t i = row * DECL_BINS + bin;
        atomicAdd(&dh[i >> 1], 1u << (16u * (i & 1u)));
    };
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
    const uint32_t m = wide ? lane & 15u : lane & 7u, cps = wide ? 64u : 32u;
    for (uint32_t sidx = wave; sidx < nb; sidx += nw) {
        const uint32_t *src = decl_b + (uint64_t)sidx * ((1u + nsl) * 64u) + lane;
        const uint32_t P = src[0];
        const bool isq = (P >> 31) != 0, act = (P & FZ_P_ACT) != 0;
        const uint32_t n = P & 0x1FFu;
        uint32_t inv = 0, hasn = 0;
        for (uint32_t u = 0; u < nsl; ++u) {
            const uint32_t w = src[64u * (1u + u)];
            for (uint32_t j = 0; j < 4; ++j) {
                const uint32_t col = cps * u + 4u * m + j;
                if (!act || col >= n || col >= lc) continue;   // (n <= lmax: longer lines are listed, not batched)
                const uint32_t b = (w >> (8u * j)) & 0xFFu;
                const bool mine = col - row0 < rows;
                if (isq) {
                    if (mine) count(col - row0, 8u + b);
                } else {
                    const uint32_t c = base_class(b);
                    inv |= c == 5 ? 1u : 0u;
                    hasn |= c == 4 ? 1u : 0u;
                    if (mine) count(col - row0, c);
                }
            }
        }
        const unsigned long long bi = __ballot(inv != 0), bn = __ballot(hasn != 0);
        if (m == 0 && act && !isq && row0 == 0) {
            const unsigned long long gm = wide ? 0xFFFFull : 0xFFull;   // (the lanes of this line)
            const bool li = ((bi >> lane) & gm) != 0, ln = ((bn >> lane) & gm) != 0;
            if (li || ln) atomicAdd(&scalars[3], ~0ull);
            if (li) atomicAdd(&scalars[4], ~0ull);
        }
    }
    for (uint32_t sidx = wave; sidx < nl; sidx += nw) {
        const uint32_t *e = reinterpret_cast<const uint32_t *>(decl_l) + 4u * sidx;
        // the line's first byte: tile, position y in the wave's data area while group g of the tile was in it (y = FZ_TAIL is the group's first byte)
        const uint64_t gs = ((uint64_t)e[0] << WT_SHIFT) + (e[1] >> 16) * FZ_GROUP + (e[1] & 0xFFFFu) - FZ_TAIL;
        const uint32_t n = e[2];
        const bool isq = e[3] != 0;
        unsigned long long over = 0;
        uint32_t inv = 0, hasn = 0;
        for (uint32_t col = lane; col < n; col += 64) {
            const uint32_t b = buf[gs + col];
            uint32_t bin = 8u + b;
            if (!isq) {
                bin = base_class(b);
                inv |= bin == 5 ? 1u : 0u;
                hasn |= bin == 4 ? 1u : 0u;
            }
            if (col >= lmax) ++over;
            else if (col - row0 < rows) count(col - row0, bin);
            else if (col >= lc && row0 == 0) {   // (lmax 512: the one row beyond the packed instance's 511)
                if (bin < 8) atomicAdd(&base_hist[(uint64_t)col * 8 + bin], 1ull);
                else atomicAdd(&qual_hist[(uint64_t)col * 256 + (bin - 8)], 1ull);
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) over += __shfl_xor(over, d);
        const bool li = __ballot(inv != 0) != 0, ln = __ballot(hasn != 0) != 0;
        if (lane == 0 && row0 == 0) {
            if (over) atomicAdd(&scalars[isq ? 6 : 5], over);
            if (!isq && (li || ln)) atomicAdd(&scalars[3], ~0ull);
            if (!isq && li) atomicAdd(&scalars[4], ~0ull);
        }
    }
    __syncthreads();
    unsigned long long over_s = 0, over_q = 0;   // (rows the pass kept beyond the caller's: dumped batches of reads longer than lmax)
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) {
        const uint32_t v = dh[i];
        if (!v) continue;
#pragma unroll
        for (uint32_t h = 0; h < 2; ++h) {
            const uint32_t c = (v >> (16u * h)) & 0xFFFFu;
            if (!c) continue;
            const uint32_t k = 2u * i + h, rw = k / DECL_BINS, bin = k - rw * DECL_BINS, row = row0 + rw;
            if (row >= lmax) {
                if (bin < 8) over_s += c;
                else over_q += c;
            } else if (bin < 8) atomicAdd(&base_hist[(uint64_t)row * 8 + bin], (unsigned long long)c);
            else atomicAdd(&qual_hist[(uint64_t)row * 256 + (bin - 8)], (unsigned long long)c);
        }
    }
    if (over_s) atomicAdd(&scalars[5], over_s);
    if (over_q) atomicAdd(&scalars[6], over_q);
}
#ifndef FQH_FZ_WP
#define FQH_FZ_WP 12
#endif
#ifndef FQH_FZ_W6
#define FQH_FZ_W6 16
#endif
#ifndef FQH_FZ_W7
#define FQH_FZ_W7 12   // wavefronts of the seven-step wide instance for rows 385 .. 448 (0: those rows take <8,12,wide>; 14: measured, slower)
#endif
#ifndef FQH_FZ_WIDE_FROM
#define FQH_FZ_WIDE_FROM 160   // rows above this take a wide instance (sixteen lanes per line, packed counters); 256: rows 161 .. 256 through <8,12>
#endif
constexpr uint32_t FZ_LC_MAX = 511;   // rows of the packed instance: a line's length travels in nine bits, and the kept tail holds 511 bytes
static uint32_t fz_lc(uint32_t rows) { return rows < FZ_LC_MAX ? rows : FZ_LC_MAX; }   // rows: FusedArgs::rows (scan_stats_rows)
// which rows take a wide (packed, sixteen lanes per line) instance
static bool fz_is_wide(uint32_t lc) { return lc > FQH_FZ_WIDE_FROM; }
static size_t stats_declined_lds(uint32_t rows) { return (size_t)std::min<uint32_t>(fz_lc(rows), SO_LC_MAX) * DECL_BINS * 2; }
// The one thing about k_stats_declined that can fail, done BEFORE the single pass is enqueued: an error behind k_stats_commit
// would leave the dumped batches and listed lines uncounted in a result that says it is complete.
hipError_t prepare_stats_declined(uint32_t rows) {
    static LdsAttr attr;
    return attr.ensure(reinterpret_cast<const void *>(k_stats_declined), stats_declined_lds(rows));
}
uint32_t scan_stats_nsl(uint32_t rows) {
    const uint32_t lc = fz_lc(rows), steps = (lc + 31) / 32;
    if (fz_is_wide(lc)) {   // the wide instances' steps of 64 columns
        const uint32_t ws = (lc + 63) / 64;
        return ws <= 3 ? 3u : ws <= 6 ? ws : (FQH_FZ_W7 && ws == 7) ? 7u : 8u;
    }
    return steps <= 2 ? 2u : steps <= 5 ? steps : 8u;
}
void launch_stats_declined(hipStream_t s, const DevOut *out, const FusedArgs &z, unsigned long long *qual_hist,
                           unsigned long long *base_hist, unsigned long long *scalars) {
    if (!z.decl_cap) return;
    const uint32_t lc = fz_lc(z.rows);   // (the rows the caller's arrays and the single pass share)
    const uint32_t nsl = scan_stats_nsl(z.rows);
    // a block's 16-bit counters hold the lines of its share of the slots: 8 lines per dumped batch, 1 per listed line
    const uint64_t per_block = DECL_LINES_PER_BLOCK / 9;
    const uint32_t blocks = (uint32_t)std::max<uint64_t>(512, ((uint64_t)z.decl_cap + per_block - 1) / per_block);
    for (uint32_t row0 = 0; row0 < lc; row0 += SO_LC_MAX) {   // (windows of 256 rows: what the LDS holds)
        const uint32_t rows = std::min<uint32_t>(lc - row0, SO_LC_MAX);
        hipLaunchKernelGGL(k_stats_declined, dim3(blocks), dim3(1024), (size_t)rows * DECL_BINS * 2, s, out, z.decl_b, nsl, z.decl_l, z.decl_cap,
                           z.buf, z.lmax, lc, row0, rows, fz_is_wide(lc) ? 1u : 0u, qual_hist, base_hist, scalars);
    }
}

uint32_t stats_blocks(int n_cu);

// The rows the single pass keeps.  An instance issues all its steps for every batch of lines, so it should fit the READS, not the
// caller's arrays: lmax is the caller's choice and may be far above the reads' length (one tool, lmax = 1000, whatever comes) —
// 150-base reads with lmax = 512 ran through the eight-step wide instance at 1 186 GB/s, with lmax = 1000 through two passes at
// 1 491, against 2 233 with lmax = 150.  hint: the longest line the caller of this function knows of in this kind of input
// (fqh_ctx::rows_hint: a look at the input's first 64 KiB, then what the context's calls found), 0: nothing known.  The rows
// are the capacity of the instance that holds the hint, at most lmax; a line beyond them is listed and counted behind the pass
// like any line beyond lmax (k_stats_declined adds its columns below lmax to the caller's arrays), so a hint that is too small
// costs time, never a count.
uint32_t scan_stats_rows(uint32_t lmax, uint32_t hint) {
    if (!hint || hint > FZ_LC_MAX) return lmax < FZ_LC_MAX ? lmax : FZ_LC_MAX;
    // (the capacity of the instance that holds the hint — also when that is MORE than lmax: reads longer than the caller's rows are
    // counted in rows of their own and k_stats_commit turns what lies beyond lmax into the overflow counters, instead of every
    // line being listed for k_stats_declined until the pass is given up)
    return hint <= FQH_FZ_WIDE_FROM ? (hint <= 64 ? 64u : (hint + 31u) / 32u * 32u) : std::min<uint32_t>(hint <= 192 ? 192u : (hint + 63u) / 64u * 64u, FZ_LC_MAX);
}
// can the single-pass kernel take this call?  (up to 160 rows of 32-bit counters, up to 511 of packed 16-bit ones; a line beyond
// the rows is listed or declined.)  By what is known of the reads: if no line is longer than 511 bytes the pass fits whatever lmax
// is (more rows than that in the caller's arrays are fine: nothing will be counted there); if MOST lines are (mostly_long:
// kilobase reads — a line that began before the 512 bytes a wavefront keeps of the previous group gives the pass up) it never
// does, and the call goes to the two-pass route without an attempt; otherwise, and with nothing known, by lmax (a few lines
// beyond the rows are listed and counted behind the pass).
bool scan_stats_supports(uint32_t lmax, uint32_t hint, bool mostly_long) {
    return lmax >= 1 && !mostly_long && ((hint && hint <= FZ_LC_MAX) || lmax <= 512);
}
uint32_t scan_stats_blocks(uint64_t n_tiles, int n_cu) {
    const uint64_t want = ((n_tiles + FZ_SPAN - 1) / FZ_SPAN + FZ_WAVES_MAX - 1) / FZ_WAVES_MAX;
    const uint32_t cus = stats_blocks(n_cu);
    return (uint32_t)(want < cus ? (want ? want : 1) : cus);
}
size_t scan_stats_scratch_bytes(int n_cu) { return (size_t)stats_blocks(n_cu) * 2 * SO_WORDS * sizeof(uint32_t); }   // (the packed instance's rows: two per LDS word)

template <uint32_t NSL, uint32_t FZ_WAVES, bool PACK = false>
static hipError_t launch_scan_stats_n(hipStream_t s, FusedArgs z, uint32_t blocks) {
    constexpr uint32_t HNSL = NSL;   // (wide + packed: eight steps of 64 columns in the geometry of eight steps of 32)
    z.wave_base = SO_SBYTES + ((HNSL + 1) / 2) * 16384u;
    const size_t lds = (size_t)z.wave_base + (size_t)FZ_WAVES * FZ_WAVE_BYTES + FZ_SLACK;
    static_assert(SO_SBYTES + ((HNSL + 1) / 2) * 16384u + FZ_WAVES * FZ_WAVE_BYTES + FZ_SLACK <= SO_LDS_MAX, "LDS budget");
    // lanes without a whole dword subtract 0 at the address their bytes form (any bin byte plus the largest
    // row-block offset): inside the allocation, and harmless wherever it lands (stats_dev.h)
    static_assert(65536 + SO_SBYTES + 128 + ((HNSL - 1) / 2) * 16384u <= SO_SBYTES + ((HNSL + 1) / 2) * 16384u + FZ_WAVES * FZ_WAVE_BYTES,
                  "garbage addresses must stay inside the allocation");
    static_assert(!PACK || (SO_SBYTES + ((HNSL + 1) / 2) * 16384u) / 4 <= SO_WORDS, "a packed instance flushes at most SO_WORDS words");
    static LdsAttr attr;
    if (hipError_t e = attr.ensure(reinterpret_cast<const void *>(k_scan_stats<NSL, FZ_WAVES, PACK>), lds); e != hipSuccess) return e;
    if (PACK) {   // the blocks ADD to their rows in scratch, epoch by epoch
        if (hipError_t e = hipMemsetAsync(z.scratch, 0, (size_t)blocks * 2 * SO_WORDS * sizeof(uint32_t), s); e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_scan_stats<NSL, FZ_WAVES, PACK>), dim3(blocks), dim3(FZ_WAVES * 64), lds, s, z);
    return hipSuccess;
}

// z: buf, len, n_tiles, the fast path's outputs, lmax, scratch (scan_stats_scratch_bytes), scalars = ZEROED side
// array of FQH_NSCALARS u64 (not the caller's: see k_stats_commit)
hipError_t launch_scan_stats(hipStream_t s, FusedArgs z, int n_cu) {
    if (!z.rows) z.rows = z.lmax;
    z.lc = fz_lc(z.rows);
#ifdef FQH_TUNING  // knock-out flags of the timing experiments (tools/exp_fzdbg.py); not part of the product library
    z.dbg = getenv("FQH_FZ_DBG") ? (uint32_t)atoi(getenv("FQH_FZ_DBG")) : 0u;
#else
    z.dbg = 0;
#endif
    const uint32_t blocks = scan_stats_blocks(z.n_tiles, n_cu);
    const uint32_t nsl = (z.lc + 31) / 32;
    // (short reads, VERDICT r4 item 8: an instance issues all its steps for every batch, so rows of 36 .. 128 columns get
    // instances of 2, 3 and 4 steps — at 50 bp three of the five steps of <5,16> count nothing)
    hipError_t e;
    if (fz_is_wide(z.lc)) {
        // wide instances (sixteen lanes per line, 64 columns per step, packed counters): up to 384 columns (MiSeq 2 x 300) at the
        // sixteen wavefronts of the 150 bp instance, up to 511 at twelve
        const uint32_t ws = (z.lc + 63) / 64;
        e = ws <= 3 ? launch_scan_stats_n<3, FQH_FZ_W5, true>(s, z, blocks)
          : ws == 4 ? launch_scan_stats_n<4, FQH_FZ_W5, true>(s, z, blocks)
          : ws == 5 ? launch_scan_stats_n<5, FQH_FZ_W5, true>(s, z, blocks)
          : ws == 6 ? launch_scan_stats_n<6, FQH_FZ_W6, true>(s, z, blocks)
#if FQH_FZ_W7
          : ws == 7 ? launch_scan_stats_n<7, FQH_FZ_W7, true>(s, z, blocks)
#endif
                    : launch_scan_stats_n<8, FQH_FZ_WP, true>(s, z, blocks);
    } else {
        // (short reads, VERDICT r4 item 8: an instance issues all its steps for every batch, so rows of 36 .. 128 columns get
        // instances of 2, 3 and 4 steps — at 50 bp three of the five steps of <5,16> count nothing)
        e = nsl <= 2 ? launch_scan_stats_n<2, FQH_FZ_W5>(s, z, blocks)
          : nsl == 3 ? launch_scan_stats_n<3, FQH_FZ_W5>(s, z, blocks)
          : nsl == 4 ? launch_scan_stats_n<4, FQH_FZ_W5>(s, z, blocks)
#if FQH_FZ_WIDE_FROM > 160   // (experiments only: rows 161 .. 256 through eight steps of 32 columns at twelve wavefronts)
          : nsl > 5 ? launch_scan_stats_n<8, 12>(s, z, blocks)
#endif
                    : launch_scan_stats_n<5, FQH_FZ_W5>(s, z, blocks);
    }
    if (e != hipSuccess) return e;
    return hipGetLastError();
}
void launch_stats_commit(hipStream_t s, const DevOut *out, const FusedArgs &z, uint32_t blocks,
                         unsigned long long *qual_hist, unsigned long long *base_hist, unsigned long long *scalars) {
    const uint32_t lc = fz_lc(z.rows);
    if (fz_is_wide(lc)) {
        hipLaunchKernelGGL(k_stats_commit_packed, dim3((2 * SO_WORDS + 255) / 256, (blocks + RED_GROUP - 1) / RED_GROUP), dim3(256), 0, s, out,
                           z.scratch, blocks, lc, z.lmax, z.scalars, qual_hist, base_hist, scalars);
        return;
    }
    const uint32_t words = (SO_SBYTES + (scan_stats_nsl(z.rows) + 1) / 2 * 16384u) / 4;
    hipLaunchKernelGGL(k_stats_commit, dim3((words + 255) / 256, (blocks + RED_GROUP - 1) / RED_GROUP), dim3(256), 0, s, out,
                       z.scratch, blocks, lc, z.lmax, words, z.scalars, qual_hist, base_hist, scalars);
}


// k_peek_lines — a look at an input before its first single pass: the longest SEQUENCE / QUALITY line — what the pass sizes its
// rows by when the context knows nothing yet about the reads (scan_stats_rows) — and how many of all lines are longer than the pass
// takes.  FOUR windows of up to 64 KiB, one block each: the input's first bytes and three more, a quarter of the input apart (a file
// whose first reads are shorter than the rest — sorted, or trimmed harder at the start of a run — used to cost its first call a
// whole wasted pass).  Window 0 knows its line phase (phase0: the place in its record, 0 header .. 3 quality, of the line the
// window begins in — the carry's newline count & 3), so line i of the window is line (phase0 + i) & 3 of a record; the other
// windows begin anywhere and settle their phase themselves: the one residue (mod 4) whose lines ALL begin with '@' while the lines
// two further on ALL begin with '+' is the headers' (src/records.rs:141, 155); no such residue, or several: the longest line of any
// kind (an over-estimate costs speed, never a count).  Header lines may be several times the reads' length and say nothing about
// the rows.  A guess, not a promise: whatever it says, the pass counts exactly.  A line that is still open at the end of a window
// counts with what the window holds of it; the line a later window begins in is skipped (its beginning is not in the window).
// out[3 w + 0]: the longest sequence / quality line of window w, [3 w + 1]: lines that end in it, [3 w + 2]: ... longer than 511 bytes.
constexpr uint32_t PEEK_WINDOWS = 4, PEEK_BYTES = 65536;
__global__ __launch_bounds__(1024) void k_peek_lines(const uint8_t *__restrict__ buf0, uint64_t len, uint32_t phase0, unsigned long long *__restrict__ out) {
    __shared__ int last_nl[1024];      // -> the last newline at or before the end of segment t (a running maximum), -1: none
    __shared__ int nl_before[1024];    // -> newlines up to and including segment t (a running sum)
    __shared__ int r_max[4], r_cnt[4], r_at[4], r_plus[4], n_lines, n_long;
    const uint32_t w = blockIdx.x, t = threadIdx.x, lo = t * 64u;
    const uint64_t start = w == 0 ? 0 : (len / PEEK_WINDOWS * w) & ~(uint64_t)15;
    const uint32_t n = (uint32_t)(len - start < PEEK_BYTES ? len - start : PEEK_BYTES);
    const uint8_t *__restrict__ buf = buf0 + start;
    if (t < 4) r_max[t] = r_cnt[t] = r_at[t] = r_plus[t] = 0;
    if (t == 0) n_lines = n_long = 0;
    int l = -1, nl = 0;
    for (uint32_t i = lo; i < lo + 64u && i < n; ++i)
        if (buf[i] == '\n') { l = (int)i; ++nl; }
    last_nl[t] = l;
    nl_before[t] = nl;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {   // inclusive scans: maximum of the last newline, sum of the newlines
        const int v = t >= d ? last_nl[t - d] : -1, c = t >= d ? nl_before[t - d] : 0;
        __syncthreads();
        if (v > last_nl[t]) last_nl[t] = v;
        nl_before[t] += c;
        __syncthreads();
    }
    // the lines this segment's newlines close: the first one began at `prev + 1` (in an earlier segment, perhaps), it is line
    // number `idx` of the window; residue = its place mod 4 (window 0: its place in the record)
    int prev = t ? last_nl[t - 1] : -1, idx = t ? nl_before[t - 1] : 0;
    const uint32_t ph = w == 0 ? phase0 : 0u;
    auto tally = [&](int len_, int first) {   // a line of the window: its length, the position of its first byte (-1: not in the window)
        const uint32_t r = (ph + (uint32_t)idx) & 3u;
        if (w != 0 && first < 0) return;      // (began in front of a later window: neither its length nor its first byte is known)
        atomicMax(&r_max[r], len_);
        atomicAdd(&r_cnt[r], 1);
        if (first >= 0 && first < (int)n) {
            const uint8_t b0 = buf[first];
            if (b0 == '@') atomicAdd(&r_at[r], 1);
            if (b0 == '+') atomicAdd(&r_plus[r], 1);
        }
    };
    for (uint32_t i = lo; i < lo + 64u && i < n; ++i) {
        if (buf[i] == '\n') {
            const int len_ = (int)i - prev - 1;
            tally(len_, prev < 0 ? (w == 0 ? 0 : -1) : prev + 1);
            atomicAdd(&n_lines, 1);
            if (len_ > (int)FZ_LC_MAX) atomicAdd(&n_long, 1);
            prev = (int)i;
            ++idx;
        }
    }
    if (lo < n && lo + 64u >= n) {                        // the segment that holds the window's end: the line still open there
        const int open = (int)n - prev - 1;
        if (open > 0) tally(open, prev < 0 ? (w == 0 ? 0 : -1) : prev + 1);
    }
    __syncthreads();
    if (t == 0) {
        int best = 0;
        if (w == 0) {
            best = max(r_max[1], r_max[3]);               // (line 1 of a record: sequence, line 3: quality)
        } else {
            int hdr = -1, cands = 0;
            for (int r = 0; r < 4; ++r)
                if (r_cnt[r] > 0 && r_at[r] == r_cnt[r] && r_cnt[(r + 2) & 3] > 0 && r_plus[(r + 2) & 3] == r_cnt[(r + 2) & 3]) { hdr = r; ++cands; }
            if (cands == 1) best = max(r_max[(hdr + 1) & 3], r_max[(hdr + 3) & 3]);
            else best = max(max(r_max[0], r_max[1]), max(r_max[2], r_max[3]));
        }
        out[3 * w + 0] = (unsigned long long)best;
        out[3 * w + 1] = (unsigned long long)n_lines;
        out[3 * w + 2] = (unsigned long long)n_long;
    }
}
void launch_peek_lines(hipStream_t s, const uint8_t *buf, uint64_t len, uint32_t phase0, unsigned long long *d_out) {
    // (inputs of up to four windows' bytes: the first window is the input)
    const uint32_t windows = len > (uint64_t)PEEK_WINDOWS * PEEK_BYTES ? PEEK_WINDOWS : 1u;
    hipLaunchKernelGGL(k_peek_lines, dim3(windows), dim3(1024), 0, s, buf, len, phase0 & 3u, d_out);
}
uint32_t peek_windows(uint64_t len) { return len > (uint64_t)PEEK_WINDOWS * PEEK_BYTES ? PEEK_WINDOWS : 1u; }

}  // namespace fqh
