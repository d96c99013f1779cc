// scan_dev.h — device-side helpers of the byte scan, shared by k_index_t / k_index_fast
// (scan_kernels.hip) and k_scan_stats (fused_kernels.hip), and the layout of the fast path's per-tile line.
#pragma once
#include <hip/hip_runtime.h>

#include "fqh_internal.h"

namespace fqh {

// ---------------------------------------------------------------------------------------------
// byte-scan helpers
// 0x80 in every byte of x that equals the (7-bit) pattern byte — exact, no false positives, three
// VALU ops: ((x & 0x7F..) ^ pat) in one v_bitop3; + 0x7F.. carries into bit 7 iff the low 7 bits
// differ; bit 7 of x itself must be clear as well: ~(a | x) & 0x80.. in one v_bitop3.
__device__ __forceinline__ uint32_t eq_flags(uint32_t x, uint32_t pat4) {
    const uint32_t a = ((x & 0x7F7F7F7Fu) ^ pat4) + 0x7F7F7F7Fu;
    return ~(a | x) & 0x80808080u;
}
// positional 16-bit mask: bit q set iff byte q of the 16-byte chunk equals the pattern byte.
// MASKV 0: shift/or nibble gather.  MASKV 1: v_dot4_u32_u8 gathers the four 0x80 flags of a dword
// in ONE instruction (byte weights 1,2,4,8 give nibble << 7; the next dword's weights 16..128 add
// its nibble four bits higher).
__device__ __forceinline__ uint32_t nib(uint32_t m) {
    m >>= 7;
    m |= m >> 7;
    m |= m >> 14;
    return m & 0xFu;
}
template <int MASKV>
__device__ __forceinline__ uint32_t eqmask16(const uint4 &v, uint32_t pat4) {
    if (MASKV == 0) {
        return nib(eq_flags(v.x, pat4)) | (nib(eq_flags(v.y, pat4)) << 4) |
               (nib(eq_flags(v.z, pat4)) << 8) | (nib(eq_flags(v.w, pat4)) << 12);
    } else {
        const uint32_t lo = __builtin_amdgcn_udot4(eq_flags(v.y, pat4), 0x80402010u,
                            __builtin_amdgcn_udot4(eq_flags(v.x, pat4), 0x08040201u, 0u, false), false);
        const uint32_t hi = __builtin_amdgcn_udot4(eq_flags(v.w, pat4), 0x80402010u,
                            __builtin_amdgcn_udot4(eq_flags(v.z, pat4), 0x08040201u, 0u, false), false);
        return (lo >> 7) | (hi << 1);
    }
}
// byte q (0..15) of a 16-byte chunk held in registers: pick the 8-byte half with two selects, then
// v_perm_b32 pulls the byte out (selector 0x0C = constant zero).  Written this way so the compiler
// does not turn it into an indexed vector extract through LDS.
__device__ __forceinline__ uint32_t byte_of(const uint4 &v, uint32_t q) {
    const uint32_t lo = q < 8 ? v.x : v.z;
    const uint32_t hi = q < 8 ? v.y : v.w;
    return __builtin_amdgcn_perm(hi, lo, 0x0C0C0C00u | (q & 7u));
}
// 16 bytes at buf+off; bytes at or beyond len read as 0
// The input is read exactly once: non-temporal loads keep it from displacing useful lines and are
// worth ~12 % of HBM read rate on MI355X (tools/readbw.hip: 5.8 -> 6.5 TB/s).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 load16_nt(const uint8_t *p) {
    const u32x4 r = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return make_uint4(r.x, r.y, r.z, r.w);
}
__device__ __forceinline__ uint4 load16(const uint8_t *__restrict__ buf, uint64_t off, uint64_t len) {
    if (off + 16 <= len) return load16_nt(buf + off);
    uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;  // no indexed array: it would be promoted to LDS
    if (off < len) {
        const uint32_t n = (uint32_t)(len - off);
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t b = (uint32_t)buf[off + i] << ((i & 3u) * 8u);
            if (i < 4) w0 |= b; else if (i < 8) w1 |= b; else if (i < 12) w2 |= b; else w3 |= b;
        }
    }
    return make_uint4(w0, w1, w2, w3);
}
// ---------------------------------------------------------------------------------------------
// helpers of the index kernel (k_index_t below)
// lane-1's value (lane 0 gets `first`): DPP wave_shr:1, no LDS crossbar round trip
__device__ __forceinline__ uint32_t wave_shr1(uint32_t x, uint32_t first) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)first, (int)x, 0x138, 0xF, 0xF, false);
}

// One 1 KiB piece: 16 bytes per lane, already in registers.  FULL: every byte of the piece exists.
template <bool FULL, int MASKV>
__device__ __forceinline__ void index_piece(const uint4 v, const uint64_t off, const uint64_t len,
                                            const uint32_t pbase, const uint32_t lane,
                                            uint32_t &prev, uint32_t &run,
                                            uint16_t *__restrict__ tl, const uint32_t list_cap) {
    const uint32_t M = eqmask16<MASKV>(v, 0x0A0A0A0Au);
    uint32_t LS = ((M << 1) | wave_shr1(M >> 15, prev)) & 0xFFFFu;
    if (!FULL && off + 16 > len) {  // a line start must be an existing byte
        const uint32_t nvalid = off < len ? (uint32_t)(len - off) : 0u;
        LS &= (1u << nvalid) - 1u;
    }
    prev = ((uint32_t)__builtin_amdgcn_readlane((int)M, 63)) >> 15;
    // exclusive prefix of popc(LS) over the wave: two ballot levels cover FASTQ ("\n+\n" puts two
    // line starts in one 16-byte chunk); deeper levels only for pathological input
    const uint32_t c = __popc(LS);
    const unsigned long long b1 = __ballot(c >= 1), b2 = __ballot(c >= 2);
    uint32_t pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, 0));
    pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b2, pre));
    uint32_t tot = (uint32_t)__popcll(b1) + (uint32_t)__popcll(b2);
    if (__ballot(c >= 3)) {
        for (uint32_t k = 3;; ++k) {
            const unsigned long long b = __ballot(c >= k);
            if (!b) break;
            pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, pre));
            tot += (uint32_t)__popcll(b);
        }
    }
    if (run + tot <= list_cap) {  // uniform: no per-entry bound check
        uint16_t *__restrict__ dst = tl + run + pre;
        while (LS) {
            const uint32_t q = __ffs(LS) - 1;
            LS &= LS - 1;
            const uint32_t b = byte_of(v, q);
            *dst++ = (uint16_t)((pbase + q) | ((b == '@') ? 0x4000u : 0u) | ((b == '+') ? 0x8000u : 0u));
        }
    }
    run += tot;
}

// fast path: one 128-byte line of u16 per tile: [0..FR_N) offsets of the tile's first record starts
// (unused slots 0), [FR_EDGE..+8) its first four and last four entries, [FR_CNT..+2) the entry count,
// [FR_HYP] the alignment (7: none).  Record starts FR_N .. FR_N + 63 of a tile (reads shorter than ~140 bp)
// go to a second whole line, fast_rs + FR2_OFF(n_tiles) + tile * 64; anything beyond into list[tile][8 + j].
// Line starts per tile the fast path can stage (reads down to ~25 bp); 6 blocks per CU fit with this, which
// measures the same as the 7 that 512 entries allow (tools/exp_ab_env.py FQH_INDEX_BPC 0 6).
constexpr uint32_t FAST_ENTRIES = 1024;
constexpr uint32_t FR_N = 52, FR_EDGE = 52, FR_CNT = 60, FR_HYP = 62, FR_STRIDE = 64, FR2_N = 64;
// [FR_HYP] >= FR_SMALL: the short partial tile at the end of the buffer (fewer than eight line starts — too few for the
// windows to single out an alignment).  Its entries 0 .. count-1 sit in the eight edge slots in order, nothing of it
// was validated or emitted by the per-tile kernels: k_finalize_fast does both, with the true line index in hand.
// Low bits: 0..3 = the alignment k_scan_stats counted the tile's lines under (finalize checks it), 4 = nothing counted.
constexpr uint32_t FR_SMALL = 8;
__host__ __device__ __forceinline__ uint64_t fr2_off(uint64_t n_tiles) { return (n_tiles + 64) * FR_STRIDE; }

}  // namespace fqh
