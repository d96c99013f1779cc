// stats_dev.h — device-side pieces of the eight-lanes-per-line histogram count, shared by
// k_stats_oct (stats_kernels.hip: lines fetched from HBM over a full tile index) and k_scan_stats
// (fused_kernels.hip: lines read back from the LDS image of the byte scan).
#pragma once
#include <hip/hip_runtime.h>

#include "fqh_internal.h"

namespace fqh {

__device__ __forceinline__ uint32_t base_class(uint32_t c) {
    return c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : c == 'N' ? 4u : 5u;
}

// One lane per record.  LDS holds u32 histograms for columns < lc (quality window 64 bins, 8 base

__device__ __forceinline__ uint32_t bin_to_class(uint32_t bin) {  // A0 C1 G2 T3 N4 other5
    return bin == 1 ? 0u : bin == 3 ? 1u : bin == 7 ? 2u : bin == 4 ? 3u : bin == 6 ? 4u : 5u;
}

constexpr uint32_t RED_GROUP = 32;  // per-block partial histograms summed by one thread of the reduce kernel

// k_stats_oct — eight lanes per line, conflict-free LDS atomics.
//
// What bounds a histogram of random bytes on a CU is the LDS atomic unit and the instruction issue
// around it.  A ds_add_u32 costs 4 LDS cycles per wave when its 2 x 32 lanes hit 32 distinct banks
// and N x that with N-way bank or address collisions (tools/ldsatom.hip; binned instrument
// qualities give 9x).  Here the bank is a function of the LANE only, so no input can collide:
//   * a line is walked by 8 consecutive lanes, one dword (4 columns) each, 32 columns per step;
//     a wave walks 8 lines at once (a "batch");
//   * the histogram is bin-major: byte address = region | rb << (8 + binbits) | bin << 8 | slot << 2
//     with rb = row / 64 and slot = a 6-bit rearrangement of row % 64 (so_slot);
//   * at the k-th atomic of a step, lane (line slot g, dword m) adds the byte j = k ^ (g & 3) of
//     its dword: row = 32 u + 4 m + j, slot = m + 8 j + 32 (u & 1), bank = m + 8 j — the 32 lanes
//     of a group (4 line slots x 8 dwords) are on 32 distinct banks whatever the bins are;
//   * the bin sits in byte 1 of the address, so one v_perm_b32 (byte 0 from the lane's register of slot
//     offsets, byte 1 from the bins, bytes 2-3 zero; row block and region in the ds immediate offset) is the
//     whole address computation.
// A wave stages its tile's line-start list in LDS.  64 lines at a time, one lane per line works out
// where the line starts and how long it is (whether it ends in '\r' only once the wave has met a CRLF);
// batches then pick that up with ds_bpermute, a batch early.  Batches alternate between the sequence and
// the quality lines of the same records.  The five loads of batch b+1 (every step of the line at once,
// unconditional) are in flight while batch b is counted.  Whole dwords of in-window bytes cost 1 VALU +
// 1 DS per byte; a line's last 1-3 columns are counted by the lane that holds that dword, under byte
// masks; bytes outside the window / alphabet and columns beyond the LDS rows take the exact per-byte
// path.  Quality bins: byte - 33 (0..63); sequence bins: byte & 7.
constexpr uint32_t SO_THREADS = 1024;
constexpr uint32_t SO_WAVES = SO_THREADS / 64;
constexpr uint32_t SO_LC_MAX = 256;           // rows kept in LDS
constexpr uint32_t SO_QBYTES = 4 * 64 * 256;  // quality region: 4 row blocks x 64 bins x 64 slots x 4 B
constexpr uint32_t SO_SBYTES = 4 * 8 * 256;   // sequence region
constexpr uint32_t SO_WORDS = (SO_QBYTES + SO_SBYTES) / 4;  // the sequence region comes first: [0, SO_SBYTES)
constexpr uint32_t SO_LISTW = 256;            // list entries per tile slot in LDS (u16 each), 256-column variant; the rest is read from memory
constexpr uint32_t SO_LISTW_REG = 512;        // ... staged per wave by the other variants
constexpr uint32_t SO_SLOT_BYTES = 2 * SO_LISTW + 256;  // a tile's slot in LDS: its list entries and 64 words of bookkeeping
constexpr uint32_t SO_LDS_MAX = 160 * 1024;
constexpr uint32_t SO_ADDR_SPAN = 65536 + SO_SBYTES + 128 + 3 * 16384;  // see launch_stats_oct
// k_stats_long keeps 128 quality bins per column ('!' .. 0xA0: Q0 .. Q127 covers PacBio HiFi's '~' = Q93, which the 64-bin window
// sent to the caller's arrays in device memory byte by byte): 4 row blocks x 128 bins x 64 slots x 4 B behind the sequence region
constexpr uint32_t SO_LQBITS = 7;
constexpr uint32_t SO_LRB = 64u * 4u << SO_LQBITS;                      // bytes of one quality row block (64 columns)
constexpr uint32_t SO_LWORDS = (SO_SBYTES + 4 * SO_LRB) / 4;
constexpr uint32_t SO_LADDR_SPAN = SO_SBYTES + 4 * SO_LRB + 128;        // bins are clamped to 7 bits before they become addresses
static_assert(SO_LADDR_SPAN <= SO_LDS_MAX, "k_stats_long: LDS budget");

__device__ __forceinline__ uint32_t so_slot(uint32_t r) {  // r = row % 64
    return ((r >> 2) & 7u) | ((r & 3u) << 3) | (r & 32u);
}
__device__ __forceinline__ uint32_t so_row6(uint32_t slot) {
    return ((slot & 7u) << 2) | ((slot >> 3) & 3u) | (slot & 32u);
}
// k_stats_long's lanes hold EIGHT contiguous columns (two registers, h = 0 / 1): row % 64 = 8 m + 4 h + j.  Its slot is
// m + 8 j + 32 h — the half h is the instruction's immediate, so the 32 lanes of a lane group (four line slots: four values of
// j; eight lanes: m) sit on the 32 banks m + 8 j.  (Round 4 used so_slot for these rows, which puts bit 2 of m into slot bit 5:
// lanes m and m + 4 on ONE bank at different addresses, a two-way conflict in every atomic — SQ_LDS_BANK_CONFLICT was exactly
// half of SQ_LDS_IDX_ACTIVE, whatever the data.)
__device__ __forceinline__ uint32_t so_slot_long(uint32_t r) {  // r = row % 64
    return ((r >> 3) & 7u) | ((r & 3u) << 3) | ((r & 4u) << 3);
}
__device__ __forceinline__ uint32_t so_row6_long(uint32_t slot) {
    return ((slot & 7u) << 3) | ((slot >> 3) & 3u) | ((slot >> 3) & 4u);
}
// word index of (bin, row): quality bins 0 .. 2^QBITS - 1 (6: the window '!'..'`' of every kernel but k_stats_long, 7: '!'..0xA0,
// which holds PacBio HiFi's '~'), sequence bins 0..7.  LONGMAP: k_stats_long's slot order
template <bool IS_SEQ, uint32_t QBITS = 6, bool LONGMAP = false>
__device__ __forceinline__ uint32_t so_word(uint32_t bin, uint32_t row) {
    const uint32_t rb = row >> 6, slot = LONGMAP ? so_slot_long(row & 63u) : so_slot(row & 63u);
    return IS_SEQ ? ((rb << 9) | (bin << 6) | slot) : SO_SBYTES / 4 + ((rb << (6 + QBITS)) | (bin << 6) | slot);
}
__device__ __forceinline__ uint32_t load4_any(const uint8_t *__restrict__ p, const uint8_t *__restrict__ end) {
    uint32_t v = 0;
    if (p + 4 <= end) {
        __builtin_memcpy(&v, p, 4);
    } else {
        for (uint32_t i = 0; p + i < end && i < 4; ++i) v |= (uint32_t)p[i] << (i * 8u);
    }
    return v;
}
__device__ __forceinline__ uint32_t load4_fast(const uint8_t *__restrict__ p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

struct SoLane {          // per-lane constants of the bank schedule
    uint32_t sel[4];     // v_perm selector of the k-th atomic: byte 0 = byte k of `slots`, byte 1 = byte
                         // j = k ^ (g & 3) of the bins, bytes 2-3 zero
    uint32_t slots;      // byte k: 4 * (m + 8 j), the slot's byte offset in a bin's 256 bytes
};

// ds_add_u32 with the u & 1 half of the slot (128 bytes) as the instruction's immediate offset.
// No return value; the kernel waits for lgkmcnt(0) before the barrier that precedes the read-out.
template <uint32_t OFF>
__device__ __forceinline__ void lds_add(uint32_t byte_addr, uint32_t v) {
    asm volatile("ds_add_u32 %0, %1 offset:%2" ::"v"(byte_addr), "v"(v), "n"(OFF));
}

// The exact per-byte statement: columns pos .. of a line of `len` columns held in w, all relative to the pass's first
// column a.col0.  Only the pass's own columns (< a.lc) are counted: what lies beyond them belongs to a later pass or
// to the overflow counters, which are plain arithmetic on the line's length; the alphabet flags cover every byte.
// (lc is the tile's view of the bank-scheduled rows: 0 in tiles that take the exact path for everything; columns the
// LDS rows do not take — and quality bytes outside '!'..'`' — go to the caller's arrays.)
template <bool IS_SEQ, uint32_t QBITS = 6, bool LONGMAP = false>
__device__ __forceinline__ void so_exact_step(const StatsArgs &a, uint32_t w, uint32_t pos, uint32_t len, uint32_t lc,
                                              uint32_t *hist, uint32_t &any_n, uint32_t &any_inv) {
    const int rem = (int)len - (int)pos;
    const uint32_t nb = rem >= 4 ? 4u : (uint32_t)(rem > 0 ? rem : 0);
    for (uint32_t j = 0; j < nb; ++j) {
        const uint32_t b = (w >> (8 * j)) & 0xFFu;
        const uint32_t col = pos + j;
        if (IS_SEQ) {
            const bool valid = b == 'A' || b == 'C' || b == 'G' || b == 'T' || b == 'N';
            const uint32_t bin = valid ? (b & 7u) : 0u;
            any_inv |= valid ? 0u : 1u;
            any_n |= b == 'N' ? 1u : 0u;
            if (col >= a.lc) continue;
            if (col < lc) atomicAdd(hist + so_word<true, 6, LONGMAP>(bin, col), 1u);
            else atomicAdd(&a.base_hist[(uint64_t)(a.col0 + col) * 8 + bin_to_class(bin)], 1ull);
        } else {
            if (col >= a.lc) continue;
            if (col < lc && b - 33u < (1u << QBITS)) atomicAdd(hist + so_word<false, QBITS, LONGMAP>(b - 33u, col), 1u);
            else atomicAdd(&a.qual_hist[(uint64_t)(a.col0 + col) * 256 + b], 1ull);
        }
    }
}

// What one lane knows about one line (worked out by one lane per line, 64 lines at a time).
constexpr uint32_t SO_P_NBT = 9, SO_P_LONG = 11, SO_P_ACT = 12, SO_P_SREL = 16;
__device__ __forceinline__ uint32_t so_pack(uint32_t s_rel, uint32_t len, uint32_t lc) {
    const uint32_t lenc = len <= lc ? len : (lc & ~3u);        // columns the whole-dword steps and the tail cover
    return (lenc & ~3u) | ((lenc & 3u) << SO_P_NBT) | ((len > lc ? 1u : 0u) << SO_P_LONG) | (1u << SO_P_ACT) |
           (s_rel << SO_P_SREL);
}

template <uint32_t NSL>
struct SoBatch {                 // one batch in flight: 8 lines, this lane's dword of each step
    uint32_t P;                  // so_pack() of this lane's line (0: no line in this slot)
    uint32_t w[NSL];             // (no load for the line's partial last dword: it is one of these, and the lane that
                                 // holds it counts its one to three bytes under byte masks -- a sixth load per batch
                                 // cost the L1 as much as any of the five)
};

struct SoAcc {                   // per-lane totals (the lane that owns a line adds it)
    uint32_t rec;
    unsigned long long bases, qual;
};
// Wave-uniform per-wave totals that need no vector registers.
struct SoTotals {
    uint32_t not_dna;            // sequence lines with an 'N' or a byte outside the alphabet
    uint32_t not_dnan;           // sequence lines with a byte outside the alphabet
    unsigned long long over_s, over_q;  // sequence / quality columns at or beyond the caller's lmax
};
// (rare: the caller asked for fewer rows than its reads are long) adds up, over the wave, what each lane's line has
// beyond lmax columns
__device__ __forceinline__ unsigned long long so_over(uint32_t len, bool has, uint32_t lmax) {
    uint32_t v = (has && len > lmax) ? len - lmax : 0u;
    if (__ballot(v != 0) == 0) return 0;
    unsigned long long t = v;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d);
    return (unsigned long long)__builtin_amdgcn_readfirstlane((int)(uint32_t)t) |
           ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(t >> 32)) << 32);
}
__device__ __forceinline__ uint32_t so_groups(unsigned long long lanes) {  // 8-lane groups with a lane set
    lanes |= lanes >> 4;
    lanes |= lanes >> 2;
    lanes |= lanes >> 1;
    return (uint32_t)__builtin_popcountll(lanes & 0x0101010101010101ull);
}

// What a lane derives from the shape of its line (whole dwords, partial tail, longer than the LDS rows)
// and its place in the group; kept across batches and worked out again only when a line of another
// shape turns up (reads of one length: once per tile kind).
template <uint32_t NSL>
struct SoShape {
    uint32_t key;             // low 16 bits of the P it was derived from
    uint32_t full[NSL];       // ~0 where this lane has a whole dword of its line at step u
    uint32_t tu;              // the step that holds the line's partial last dword (column nfull4) ...
    uint32_t tb;              // ... the byte mask of the lane that holds it (0xFF per byte of the line; 0 in the other lanes) ...
    uint32_t tf[4];           // ... and the value of that lane's k-th atomic there (~0: the byte k ^ (g & 3) counts)
    uint32_t any;             // wave-uniform: 1 some line has a partial last dword, 2 some line is longer than the rows,
                              // bits 8-15: the step of the partial last dwords if it is the same for all of them, else 0xFF
};
template <uint32_t NSL>
__device__ __forceinline__ void so_shape(SoShape<NSL> &S, uint32_t P, uint32_t m) {
    S.key = P & 0xFFFFu;
    const uint32_t nfull4 = P & 0x1FFu;
    const int tt = (int)nfull4 - (int)(4u * m);
#pragma unroll
    for (uint32_t u = 0; u < NSL; ++u) S.full[u] = tt > (int)(32u * u) ? 0xFFFFFFFFu : 0u;
    const uint32_t nbt = (P >> SO_P_NBT) & 3u;
    // In step tu lane mt = nfull4 / 4 % 8 holds columns nfull4 .. nfull4 + 3, of which nbt belong to the line;
    // the lanes below it hold whole dwords, the lanes above it nothing.
    const uint32_t tu = nfull4 >> 5, mt = (nfull4 >> 2) & 7u, g3 = (__lane_id() >> 3) & 3u;
    S.tu = nbt ? tu : 7u;  // (7: no partial dword here; the masks below are 0 then)
    S.tb = (nbt && m == mt) ? (1u << (8u * nbt)) - 1u : 0u;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) S.tf[k] = (m == mt && (k ^ g3) < nbt) ? 0xFFFFFFFFu : 0u;
    const unsigned long long tl = __ballot(nbt != 0);
    uint32_t tus = 0xFFu;
    if (tl) {
        const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)tu, (int)(__ffsll((long long)tl) - 1));
        if (__ballot(nbt != 0 && tu != t0) == 0) tus = t0;
    }
    S.any = (tl ? 1u : 0u) | (__ballot((P >> SO_P_LONG) & 1u) != 0 ? 2u : 0u) | (tus << 8);
}
// ds_sub_u32 of a lane mask (~0 counts one, 0 counts nothing) with the row block and the u & 1 half of
// the slot as the instruction's immediate offset.
// (A builtin atomic on an LDS address, not inline asm: the compiler then counts these in its s_waitcnt lgkmcnt(N)
// and a wave that waits for a ds_bpermute issued before them does not wait for them as well.)
typedef __attribute__((address_space(3))) uint32_t so_lds_u32;
template <uint32_t OFF>
__device__ __forceinline__ void lds_sub(uint32_t byte_addr, uint32_t v) {
    (void)__hip_atomic_fetch_sub((so_lds_u32 *)(uintptr_t)(byte_addr + OFF), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Count one batch (its loads were issued one batch earlier), straight-line: pass 1 checks every byte
// the batch counts (whole dwords under the lane's masks, the partial tail over filler bytes), pass 2
// adds them -- one v_perm_b32 and one ds_sub per byte; lanes without a whole dword subtract 0 at
// whatever address their bytes give (the kernel's LDS allocation covers every address a byte can
// form).  A byte outside the window / alphabet sends the whole batch to the exact path instead.
template <bool IS_SEQ, uint32_t NSL, bool DBG>
__device__ __forceinline__ void so_count(const StatsArgs &a, const uint8_t *tbase, SoBatch<NSL> &B, SoShape<NSL> &S,
                                         uint32_t lane, uint32_t lc, uint32_t *hist, const SoLane &c, uint32_t my_len,
                                         uint32_t src4, SoTotals &T, SoAcc &acc, bool trimmed, bool &cr_seen, long long lrec) {
    const uint32_t m = lane & 7u, m4 = m * 4u;
    const uint32_t P = B.P;
    if (__ballot((P & 0xFFFFu) != S.key) != 0) so_shape<NSL>(S, P, m);
    const uint32_t nfull4 = P & 0x1FFu;                            // columns covered by whole dwords
    constexpr uint32_t RB = IS_SEQ ? 2048u : 16384u;               // address step of a row block
    constexpr uint32_t REGION = IS_SEQ ? 0u : SO_SBYTES;           // the region's base goes into the immediate offset too
    const uint32_t any = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.any);
    const bool tails = (any & 1u) != 0, longs = (any & 2u) != 0;
    uint32_t chk = 0;  // sequence: OR of (dword ^ expected); quality: OR of (byte - 33), bits 6-7 tell
    uint32_t orw = 0;  // sequence: OR of the counted bytes; bit 3 is set in 'N' only
    // The partial last dwords: the raw dword of step tu (one scalar pick when every line has it in the same step),
    // checked under the byte mask of the one lane that holds it, counted by that lane after pass 2.
    uint32_t pt = 0;
    const uint32_t tus = (any >> 8) & 0xFFu;
    if (tails) {
        uint32_t x = B.w[0];
        if (tus < NSL) {
#pragma unroll
            for (uint32_t u = 1; u < NSL; ++u)
                if (tus == u) x = B.w[u];
        } else {
#pragma unroll
            for (uint32_t u = 1; u < NSL; ++u) x = S.tu == u ? B.w[u] : x;
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
#define FQH_SO_PASS1(U)                                                                            \
    if (U < NSL) {                                                                                 \
        const uint32_t w = B.w[U < NSL ? U : 0], f = S.full[U < NSL ? U : 0];                      \
        if (IS_SEQ) {                                                                              \
            const uint32_t bins = w & 0x07070707u;                                                 \
            chk |= (w ^ __builtin_amdgcn_perm(0x474EFF54u, 0x43FF41FFu, bins)) & f;                \
            orw |= w & f;                                                                          \
            B.w[U < NSL ? U : 0] = bins;                                                           \
        } else {                                                                                   \
            /* byte - 33 < 64 for all four bytes: a byte below '!' borrows, but its own         */ \
            /* difference is then >= 0xDF, one above '`' gives >= 0x40: bits 6-7 tell           */ \
            const uint32_t t = w - 0x21212121u;                                                    \
            chk |= t & f;                                                                          \
            B.w[U < NSL ? U : 0] = t;                                                              \
        }                                                                                          \
    }
    FQH_SO_PASS1(0) FQH_SO_PASS1(1) FQH_SO_PASS1(2) FQH_SO_PASS1(3)
    FQH_SO_PASS1(4) FQH_SO_PASS1(5) FQH_SO_PASS1(6) FQH_SO_PASS1(7)
#undef FQH_SO_PASS1
    uint32_t slow = 0;        // wave-uniform: steps left to the exact path
    bool tail_exact = false;
    if (__ballot(IS_SEQ ? chk != 0 : (chk & 0xC0C0C0C0u) != 0) != 0) {
        slow = (1u << NSL) - 1u;
        tail_exact = tails;
    } else if (!DBG || !(a.dbg & 1u)) {
#define FQH_SO_PASS2(U)                                                                            \
        if (U < NSL) {                                                                             \
            const uint32_t pb = B.w[U < NSL ? U : 0], f = S.full[U < NSL ? U : 0];                 \
            _Pragma("unroll") for (int k = 0; k < 4; ++k)                                          \
                lds_sub<REGION + 128u * (U & 1u) + RB * (U >> 1)>(__builtin_amdgcn_perm(c.slots, pb, c.sel[k]), f); \
        }
        FQH_SO_PASS2(0) FQH_SO_PASS2(1) FQH_SO_PASS2(2) FQH_SO_PASS2(3)
        FQH_SO_PASS2(4) FQH_SO_PASS2(5) FQH_SO_PASS2(6) FQH_SO_PASS2(7)
#undef FQH_SO_PASS2
        if (tails) {  // (the row block and slot half of step tu go into the address, not the immediate offset)
            const uint32_t tu = tus < NSL ? tus : S.tu;
            const uint32_t off = REGION + ((tu & 1u) << 7) + (tu >> 1) * RB;
#pragma unroll
            for (int k = 0; k < 4; ++k) lds_sub<0>(__builtin_amdgcn_perm(c.slots, pt, c.sel[k]) + off, S.tf[k]);
        }
    }
    uint32_t any_n = IS_SEQ ? orw & 0x08080808u : 0u, any_inv = 0;
    // exact work: a refused batch (every step and the tails), and everything from column nfull4 on in
    // lines longer than the LDS rows
    if (__builtin_amdgcn_readfirstlane((int)(slow | (longs ? 512u : 0u))) != 0) {
        any_n = 0;
        const uint8_t *const bend = a.buf + a.len;
        const uint8_t *const line = tbase + (P >> SO_P_SREL);
        // (every lane takes part in the permute: a disabled source lane would read as 0)
        const uint32_t len_src = (uint32_t)__builtin_amdgcn_ds_bpermute((int)src4, (int)my_len);
        uint32_t len = (P >> SO_P_ACT) & 1u ? len_src : 0u;
        if (!trimmed) {  // (wave-uniform) the tile's lengths were taken without looking for a '\r' at the line's end: a
            // line that has one fails pass 1 (neither alphabet holds '\r') and is trimmed here; the wave looks
            // before it packs from its next tile on (trim_winline, src/records.rs:66-73)
            const bool cr = len != 0 && line[len - 1] == '\r';
            if (cr) {
                --len;
                if (m == 0) {
                    if (IS_SEQ) acc.bases -= 1;
                    else acc.qual -= 1;
                }
            }
            if (__ballot(cr) != 0) cr_seen = true;
        }
        const bool islong = ((P >> SO_P_LONG) & 1u) != 0;
        bool tail = tail_exact || longs;
        if (!slow) any_n = IS_SEQ ? orw & 0x08080808u : 0u;   // the counted part stands
        for (uint32_t ul = lc >> 5;;) {
            uint32_t pos, le;
            if (slow) {
                const uint32_t u = (uint32_t)__builtin_ctz(slow);
                slow &= slow - 1;
                pos = m4 + 32 * u;
                le = nfull4 < len ? nfull4 : len;
            } else if (tail) {
                tail = false;
                pos = nfull4;
                le = (m == 0 && (tail_exact || islong)) ? len : 0u;
            } else {
                if (!longs) break;
                pos = m4 + 32 * ul++;
                if (__ballot(pos < len) == 0) break;
                le = (islong && pos > nfull4) ? len : 0u;
            }
            const uint32_t wl = pos < le ? load4_any(line + pos, bend) : 0u;
            so_exact_step<IS_SEQ>(a, wl, pos, le, lc, hist, any_n, any_inv);
        }
    }
    if (IS_SEQ) {  // lines that are not pure ACGT / ACGTN: the 8 lanes of a line OR their flags
        const unsigned long long bi = __ballot(any_inv != 0), bn = __ballot(any_n != 0) | bi;
        if (bn) {
            if (!a.flagmap) {
                T.not_dna += so_groups(bn);
                T.not_dnan += so_groups(bi);
            } else {
                // several passes look at (different columns of) this line: its record's bit in the two maps says whether an
                // earlier one has counted it already.  lrec + slot = the record's index among those that count.
                const uint32_t sh = lane & 56u;
                const bool gn = ((bn >> sh) & 0xFFull) != 0, gi = ((bi >> sh) & 0xFFull) != 0;
                bool newn = false, newi = false;
                if (m == 0 && gn) {
                    const unsigned long long rec = (unsigned long long)(lrec + (long long)(src4 >> 2));
                    const uint32_t bit = 1u << (rec & 31u);
                    newn = !(atomicOr(&a.flagmap[rec >> 5], bit) & bit);
                    if (gi) newi = !(atomicOr(&a.flagmap[a.flag_words + (rec >> 5)], bit) & bit);
                }
                T.not_dna += (uint32_t)__popcll(__ballot(newn));
                T.not_dnan += (uint32_t)__popcll(__ballot(newi));
            }
        }
    }
}

}  // namespace fqh
