// stats_kernels.hip — per-position Phred-quality and base-composition histograms,
// the synthetic-FASTQ generator and the streaming-read ceiling probe.
//
// k_stats_records is the GPU form of "for record in records: for p: hist[p][seq()[p]] += 1"
// over the record index k_emit produced: accessors as src/records.rs:75-90 (one trailing '\r'
// trimmed), alphabets as src/records.rs:19-33.  Counters are integers: addition commutes, so the
// result is bit-exact whatever the execution order.
#include <hip/hip_runtime.h>

#include "fqh_internal.h"

namespace fqh {

constexpr uint32_t QWIN_LO = 33;   // '!' : LDS window of quality bins [33, 97)
constexpr uint32_t QWIN = 64;
constexpr uint32_t STATS_LC_MAX = 224;  // columns kept in LDS: 224 * (64 + 8) * 4 B = 63 KiB

__device__ __forceinline__ uint32_t base_class(uint32_t c) {
    return c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : c == 'N' ? 4u : 5u;
}

// One lane per record.  LDS holds u32 histograms for columns < lc (quality window 64 bins, 8 base
// classes); everything outside goes straight to the u64 global arrays.
__global__ __launch_bounds__(256) void k_stats_records(const uint8_t *__restrict__ buf,
                                                       uint64_t base_offset,
                                                       const fqh_idx_record *__restrict__ idx,
                                                       uint64_t n_records, uint32_t lmax, uint32_t lc,
                                                       unsigned long long *__restrict__ qual_hist,
                                                       unsigned long long *__restrict__ base_hist,
                                                       unsigned long long *__restrict__ scalars) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t *qh = lds;             // [lc][64]
    uint32_t *bh = lds + lc * QWIN;  // [lc][8]
    const uint32_t nlds = lc * (QWIN + 8);
    for (uint32_t i = threadIdx.x; i < nlds; i += blockDim.x) lds[i] = 0;
    __syncthreads();

    unsigned long long s_rec = 0, s_bases = 0, s_qual = 0, s_dna = 0, s_dnan = 0, s_oseq = 0, s_oqual = 0;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_records;
         k += (uint64_t)gridDim.x * blockDim.x) {
        const fqh_idx_record r = idx[k];
        const uint8_t *rec = buf + (r.start - base_offset);
        const uint8_t *seq = rec + r.head + 1;
        uint32_t sl = r.seq - r.head - 1;
        if (sl && seq[sl - 1] == '\r') --sl;  // trim_winline, src/records.rs:66-73
        const uint8_t *qual = rec + r.sep + 1;
        uint32_t ql = r.qual - r.sep - 1;
        if (ql && qual[ql - 1] == '\r') --ql;
        bool dna = true, dnan = true;
        for (uint32_t p = 0; p < sl; ++p) {
            const uint32_t c = base_class(seq[p]);
            dna &= c < 4;
            dnan &= c < 5;
            if (p < lc) atomicAdd(&bh[p * 8 + c], 1u);
            else if (p < lmax) atomicAdd(&base_hist[(uint64_t)p * 8 + c], 1ull);
            else ++s_oseq;
        }
        for (uint32_t p = 0; p < ql; ++p) {
            const uint32_t q = qual[p];
            if (p < lc && q - QWIN_LO < QWIN) atomicAdd(&qh[p * QWIN + (q - QWIN_LO)], 1u);
            else if (p < lmax) atomicAdd(&qual_hist[(uint64_t)p * 256 + q], 1ull);
            else ++s_oqual;
        }
        ++s_rec;
        s_bases += sl;
        s_qual += ql;
        s_dna += dna ? 1 : 0;
        s_dnan += dnan ? 1 : 0;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < lc * QWIN; i += blockDim.x) {
        const uint32_t v = qh[i];
        if (v) atomicAdd(&qual_hist[(uint64_t)(i / QWIN) * 256 + QWIN_LO + (i % QWIN)], (unsigned long long)v);
    }
    for (uint32_t i = threadIdx.x; i < lc * 8; i += blockDim.x) {
        const uint32_t v = bh[i];
        if (v) atomicAdd(&base_hist[i], (unsigned long long)v);
    }
    // scalars: wave reduce, one atomic per wave
    unsigned long long sc[7] = {s_rec, s_bases, s_qual, s_dna, s_dnan, s_oseq, s_oqual};
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        unsigned long long v = sc[j];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&scalars[j], v);
    }
}

void launch_stats_records(hipStream_t s, const uint8_t *buf, uint64_t base_offset,
                          const fqh_idx_record *idx, uint64_t n_records, uint32_t lmax,
                          uint64_t *qual_hist, uint64_t *base_hist, uint64_t *scalars, int n_cu) {
    if (!n_records) return;
    const uint32_t lc = lmax < STATS_LC_MAX ? lmax : STATS_LC_MAX;
    const size_t lds = (size_t)lc * (QWIN + 8) * sizeof(uint32_t);
    uint64_t blocks = (n_records + 255) / 256;
    const uint64_t maxb = (uint64_t)(n_cu > 0 ? n_cu : 256) * 4;
    if (blocks > maxb) blocks = maxb;
    hipLaunchKernelGGL(k_stats_records, dim3((uint32_t)blocks), dim3(256), lds, s, buf, base_offset, idx,
                       n_records, lmax, lc, (unsigned long long *)qual_hist,
                       (unsigned long long *)base_hist, (unsigned long long *)scalars);
}

// ---------------------------------------------------------------------------------------------
// k_stats_lines — the production histogram kernel.
//
// Unit of parallelism = one LINE per lane.  The scan's tile index already lists every line start,
// and the tile prefix gives each line its global index, hence its role (index % 4 == 1: sequence,
// == 3: quality).  A wavefront takes a 16 KiB tile, its lanes take that tile's sequence/quality
// lines (~100), and all 64 lanes walk their lines in lock step, four columns per iteration: one
// unaligned dword load per lane, a SWAR validity test, four LDS atomic adds on row p..p+3 of the
// block's histogram.  Both histograms use 64-word rows so sequence and quality lanes share the
// code: quality bin = byte - 33 (window '!'..'`'); sequence bin = byte & 7 (A1 C3 T4 N6 G7, distinct)
// replicated in 8 copies (copy = lane & 7) so that the four hot letters spread over all 32 banks.
// Anything outside the fast case (bytes outside the window / alphabet, columns beyond the LDS rows,
// the last partial dword of a line) goes through an exact per-byte path.  Counters are integers:
// the result is bit-exact whatever the order.
constexpr uint32_t SL_THREADS = 1024;
constexpr uint32_t SL_WAVES = SL_THREADS / 64;
constexpr uint32_t SL_LC_MAX = 256;  // 256 rows * 128 words * 4 B = 128 KiB of LDS

__device__ __forceinline__ uint32_t seq_expected(uint32_t w) {
    // byte-wise: the alphabet letter whose (byte & 7) equals this byte's, 0xFF where there is none
    return __builtin_amdgcn_perm(0x474EFF54u, 0x43FF41FFu, w & 0x07070707u);
}
__device__ __forceinline__ uint32_t bin_to_class(uint32_t bin) {  // A0 C1 G2 T3 N4 other5
    return bin == 1 ? 0u : bin == 3 ? 1u : bin == 7 ? 2u : bin == 4 ? 3u : bin == 6 ? 4u : 5u;
}

// 16 bytes at p (any alignment); bytes at or beyond `end` read as 0
__device__ __forceinline__ uint4 load16_any(const uint8_t *__restrict__ p, const uint8_t *__restrict__ end) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (p + 16 <= end) {
        __builtin_memcpy(&v, p, 16);
    } else {
        uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
        for (uint32_t i = 0; p + i < end && i < 16; ++i) {
            const uint32_t b = (uint32_t)p[i] << ((i & 3u) * 8u);
            if (i < 4) w0 |= b; else if (i < 8) w1 |= b; else if (i < 12) w2 |= b; else w3 |= b;
        }
        v = make_uint4(w0, w1, w2, w3);
    }
    return v;
}

// One dword (columns p..p+3) of one line per lane.  IS_SEQ is wave-uniform: a wave iteration takes
// either sequence lines or quality lines, so there is no divergence between the two alphabets.
template <bool IS_SEQ>
__device__ __forceinline__ void stats_dword(const StatsArgs &a, uint32_t w, uint32_t p, uint32_t len,
                                            uint32_t lc, uint32_t *__restrict__ row0, uint32_t *__restrict__ qh,
                                            uint32_t *__restrict__ sh, uint32_t copy8, uint32_t &any_n,
                                            uint32_t &any_inv, uint32_t &ovf) {
    if (p >= len) return;
    const uint32_t nb = len - p < 4 ? len - p : 4u;
    bool fast = nb == 4 && p + 4 <= lc;
    uint32_t bins;
    if (IS_SEQ) {
        fast = fast && w == seq_expected(w);
        bins = w & 0x07070707u;
    } else {
        const uint32_t lo7 = w & 0x7F7F7F7Fu;
        const uint32_t ge33 = lo7 + 0x5F5F5F5Fu, ge97 = lo7 + 0x1F1F1F1Fu;
        fast = fast && ((ge33 & ~ge97 & ~w) & 0x80808080u) == 0x80808080u;
        bins = w - 0x21212121u;
    }
    if (fast) {
        if (IS_SEQ) any_n |= ~((((w & 0x7F7F7F7Fu) ^ 0x4E4E4E4Eu) + 0x7F7F7F7Fu) | w) & 0x80808080u;
        uint32_t *r = row0 + p * 64;
        atomicAdd(r + (bins & 0xFFu), 1u);
        atomicAdd(r + 64 + ((bins >> 8) & 0xFFu), 1u);
        atomicAdd(r + 128 + ((bins >> 16) & 0xFFu), 1u);
        atomicAdd(r + 192 + (bins >> 24), 1u);
    } else {
        for (uint32_t j = 0; j < nb; ++j) {
            const uint32_t b = (w >> (8 * j)) & 0xFFu;
            const uint32_t col = p + j;
            if (IS_SEQ) {
                const bool valid = b == 'A' || b == 'C' || b == 'G' || b == 'T' || b == 'N';
                const uint32_t bin = valid ? (b & 7u) : 0u;
                any_inv |= valid ? 0u : 1u;
                any_n |= b == 'N' ? 1u : 0u;
                if (col < lc) atomicAdd(sh + col * 64 + copy8 + bin, 1u);
                else if (col < a.lmax) atomicAdd(&a.base_hist[(uint64_t)col * 8 + bin_to_class(bin)], 1ull);
                else ++ovf;
            } else {
                if (col < lc && b - 33u < 64u) atomicAdd(qh + col * 64 + (b - 33u), 1u);
                else if (col < a.lmax) atomicAdd(&a.qual_hist[(uint64_t)col * 256 + b], 1ull);
                else ++ovf;
            }
        }
    }
}

struct StatsAcc {
    unsigned long long rec, bases, qual, dna, dnan, oseq, oqual;
};

// All sequence lines (IS_SEQ) or all quality lines of one tile: entries i == i0 (mod 4).
template <bool IS_SEQ>
__device__ __forceinline__ void stats_tile_lines(const StatsArgs &a, uint32_t lane, uint32_t i0, uint32_t cnt,
                                                 unsigned long long lbase, uint64_t tb, uint64_t next_first,
                                                 const uint16_t *__restrict__ tl, uint32_t lc, uint32_t *qh,
                                                 uint32_t *sh, uint32_t copy8, StatsAcc &acc) {
    if (i0 >= cnt) return;
    const uint32_t nitems = (cnt - i0 + 3) >> 2;
    const uint8_t *const bend = a.buf + a.len;
    uint32_t *const row0 = IS_SEQ ? sh + copy8 : qh;
    for (uint32_t kk = 0; kk < nitems; kk += 64) {
        const uint32_t k = kk + lane;
        const uint32_t i = i0 + 4 * k;
        const unsigned long long L = lbase + i;
        const bool act = k < nitems && L >= a.line_lo && L < a.line_hi;
        uint64_t S = 0;
        uint32_t len = 0;
        if (act) {
            S = tb + (tl[i] & 0x3FFFu);
            uint64_t nextS = (i + 1 < cnt) ? tb + (tl[i + 1] & 0x3FFFu) : next_first;
            if (nextS > a.valid_end) nextS = a.valid_end;
            len = (uint32_t)(nextS - 1 - S);                      // raw line, without its '\n'
            if (len && a.buf[S + len - 1] == '\r') --len;          // trim_winline, src/records.rs:66-73
        }
        const uint8_t *__restrict__ lp = a.buf + S;
        uint32_t any_n = 0, any_inv = 0, ovf = 0;
        // longest line of this wave iteration (uniform loop bound)
        uint32_t maxlen = len;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t o = __shfl_xor(maxlen, d);
            maxlen = o > maxlen ? o : maxlen;
        }
        maxlen = (uint32_t)__builtin_amdgcn_readfirstlane((int)maxlen);
        uint4 cur = len ? load16_any(lp, bend) : make_uint4(0, 0, 0, 0);
        for (uint32_t p = 0; p < maxlen; p += 16) {
            const uint4 nxt = p + 16 < len ? load16_any(lp + p + 16, bend) : make_uint4(0, 0, 0, 0);
            const bool in = p < len;
            // Fast step (wave-uniform): every lane still inside its line has 16 more bytes, all of
            // them in the alphabet / quality window, and the 16 rows are LDS-resident.
            bool ok = len >= p + 16;
            uint4 bins;
            if (IS_SEQ) {
                ok = ok && ((cur.x ^ seq_expected(cur.x)) | (cur.y ^ seq_expected(cur.y)) |
                            (cur.z ^ seq_expected(cur.z)) | (cur.w ^ seq_expected(cur.w))) == 0;
                bins = make_uint4(cur.x & 0x07070707u, cur.y & 0x07070707u, cur.z & 0x07070707u, cur.w & 0x07070707u);
            } else {
                auto win = [](uint32_t w) {  // 0x80 per byte inside ['!', '`']
                    const uint32_t lo7 = w & 0x7F7F7F7Fu;
                    return (lo7 + 0x5F5F5F5Fu) & ~(lo7 + 0x1F1F1F1Fu) & ~w;
                };
                ok = ok && ((win(cur.x) & win(cur.y) & win(cur.z) & win(cur.w)) & 0x80808080u) == 0x80808080u;
                bins = make_uint4(cur.x - 0x21212121u, cur.y - 0x21212121u, cur.z - 0x21212121u, cur.w - 0x21212121u);
            }
            if (p + 16 <= lc && __ballot(in && !ok) == 0) {
                if (in) {
                    if (IS_SEQ) {
                        auto isn = [](uint32_t w) { return ~((((w & 0x7F7F7F7Fu) ^ 0x4E4E4E4Eu) + 0x7F7F7F7Fu) | w); };
                        any_n |= (isn(cur.x) | isn(cur.y) | isn(cur.z) | isn(cur.w)) & 0x80808080u;
                    }
                    uint32_t *r = row0 + p * 64;
#define FQH_ADD4(w, base)                                     \
                    atomicAdd(r + (base) + ((w) & 0xFFu), 1u);               \
                    atomicAdd(r + (base) + 64 + (((w) >> 8) & 0xFFu), 1u);   \
                    atomicAdd(r + (base) + 128 + (((w) >> 16) & 0xFFu), 1u); \
                    atomicAdd(r + (base) + 192 + ((w) >> 24), 1u);
                    FQH_ADD4(bins.x, 0) FQH_ADD4(bins.y, 256) FQH_ADD4(bins.z, 512) FQH_ADD4(bins.w, 768)
#undef FQH_ADD4
                }
            } else {
                stats_dword<IS_SEQ>(a, cur.x, p, len, lc, row0, qh, sh, copy8, any_n, any_inv, ovf);
                stats_dword<IS_SEQ>(a, cur.y, p + 4, len, lc, row0, qh, sh, copy8, any_n, any_inv, ovf);
                stats_dword<IS_SEQ>(a, cur.z, p + 8, len, lc, row0, qh, sh, copy8, any_n, any_inv, ovf);
                stats_dword<IS_SEQ>(a, cur.w, p + 12, len, lc, row0, qh, sh, copy8, any_n, any_inv, ovf);
            }
            cur = nxt;
        }
        if (act) {
            if (IS_SEQ) {
                ++acc.rec;
                acc.bases += len;
                acc.dna += (any_n | any_inv) ? 0 : 1;
                acc.dnan += any_inv ? 0 : 1;
                acc.oseq += ovf;
            } else {
                acc.qual += len;
                acc.oqual += ovf;
            }
        }
    }
}

__global__ __launch_bounds__(SL_THREADS) void k_stats_lines(StatsArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];  // [lc][64] quality, [lc][64] sequence
    const uint32_t lc = a.lc;
    uint32_t *const qh = hist;
    uint32_t *const sh = hist + lc * 64;
    for (uint32_t i = threadIdx.x; i < lc * 128; i += SL_THREADS) hist[i] = 0;
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = threadIdx.x >> 6;
    const uint32_t copy8 = (lane & 7u) * 8u;
    StatsAcc acc = {0, 0, 0, 0, 0, 0, 0};

    for (uint64_t tile = (uint64_t)blockIdx.x * SL_WAVES + wv; tile < a.n_tiles;
         tile += (uint64_t)gridDim.x * SL_WAVES) {
        uint32_t cnt = a.tile_count[tile];
        cnt = cnt < a.list_cap ? cnt : a.list_cap;
        if (cnt == 0) continue;
        const unsigned long long lbase = a.nl_count + 1 + a.block_prefix[tile >> SCAN_SHIFT] + a.tile_prefix[tile];
        if (lbase >= a.line_hi || lbase + cnt <= a.line_lo) continue;
        const uint16_t *__restrict__ tl = a.list + tile * a.list_cap;
        const uint64_t tb = tile << WT_SHIFT;
        // start of the first line after this tile (ends the tile's last line)
        uint64_t next_first = a.valid_end;
        for (uint64_t t2 = tile + 1; t2 < a.n_tiles; ++t2) {
            if (a.tile_count[t2]) { next_first = (t2 << WT_SHIFT) + (a.list[t2 * a.list_cap] & 0x3FFFu); break; }
        }
        const uint32_t lb3 = (uint32_t)lbase & 3u;
        stats_tile_lines<true>(a, lane, (1u - lb3) & 3u, cnt, lbase, tb, next_first, tl, lc, qh, sh, copy8, acc);
        stats_tile_lines<false>(a, lane, (3u - lb3) & 3u, cnt, lbase, tb, next_first, tl, lc, qh, sh, copy8, acc);
    }
    __syncthreads();
    uint32_t *__restrict__ dst = a.scratch + (uint64_t)blockIdx.x * lc * 128;
    for (uint32_t i = threadIdx.x; i < lc * 128; i += SL_THREADS) dst[i] = hist[i];
    unsigned long long sc[7] = {acc.rec, acc.bases, acc.qual, acc.dna, acc.dnan, acc.oseq, acc.oqual};
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        unsigned long long v = sc[j];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
        if (lane == 0 && v) atomicAdd(&a.scalars[j], v);
    }
}

// Sum the per-block partial histograms into the caller's u64 arrays.  blockIdx.y splits the partial
// histograms into groups so that no thread walks more than 32 of them; one atomic per (bin, group).
constexpr uint32_t RED_GROUP = 32;
__global__ __launch_bounds__(256) void k_stats_reduce(const uint32_t *__restrict__ scratch, uint32_t n_blocks,
                                                      uint32_t lc, unsigned long long *__restrict__ qual_hist,
                                                      unsigned long long *__restrict__ base_hist) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nq = lc * 64, ns = lc * 8;
    const uint64_t stride = (uint64_t)lc * 128;
    const uint32_t b0 = blockIdx.y * RED_GROUP;
    const uint32_t b1 = b0 + RED_GROUP < n_blocks ? b0 + RED_GROUP : n_blocks;
    if (id < nq) {
        unsigned long long s = 0;
        for (uint32_t b = b0; b < b1; ++b) s += scratch[b * stride + id];
        if (s) atomicAdd(&qual_hist[(uint64_t)(id / 64) * 256 + 33 + (id % 64)], s);
    } else if (id < nq + ns) {
        const uint32_t j = id - nq, row = j / 8, bin = j % 8;
        unsigned long long s = 0;
        for (uint32_t b = b0; b < b1; ++b)
            for (uint32_t c = 0; c < 8; ++c) s += scratch[b * stride + nq + row * 64 + c * 8 + bin];
        if (s) atomicAdd(&base_hist[(uint64_t)row * 8 + bin_to_class(bin)], s);  // bins 0,2,5 share class 5
    }
}

uint32_t stats_lines_lc(uint32_t lmax) { return lmax < SL_LC_MAX ? lmax : SL_LC_MAX; }
uint32_t stats_lines_blocks(int n_cu) { return (uint32_t)(n_cu > 0 ? n_cu : 256); }
size_t stats_lines_scratch_bytes(uint32_t lmax, int n_cu) {
    return (size_t)stats_lines_blocks(n_cu) * stats_lines_lc(lmax) * 128 * sizeof(uint32_t);
}
hipError_t launch_stats_lines(hipStream_t s, StatsArgs a, int n_cu) {
    a.lc = stats_lines_lc(a.lmax);
    const size_t lds = (size_t)a.lc * 128 * sizeof(uint32_t);
    static size_t lds_set = 0;
    if (lds > lds_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_stats_lines),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_set = lds;
    }
    const uint32_t blocks = stats_lines_blocks(n_cu);
    hipLaunchKernelGGL(k_stats_lines, dim3(blocks), dim3(SL_THREADS), lds, s, a);
    const uint32_t nred = a.lc * 72;
    hipLaunchKernelGGL(k_stats_reduce, dim3((nred + 255) / 256, (blocks + RED_GROUP - 1) / RED_GROUP), dim3(256), 0,
                       s, a.scratch, blocks, a.lc, a.qual_hist, a.base_hist);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Synthetic 150 bp FASTQ: byte b of record i is a pure function of (seed, i, b); the
// tests regenerate any sub-range on the CPU from the same map.
__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}
__device__ __forceinline__ unsigned long long synth_hash(unsigned long long seed, unsigned long long rec,
                                                         uint32_t stream, uint32_t p) {
    return mix64(seed + rec * 0x9E3779B97F4A7C15ull +
                 ((((unsigned long long)stream) << 32 | p) + 1) * 0xD6E8FEB86659FD93ull);
}
__device__ uint32_t synth_byte(unsigned long long seed, unsigned long long rec, uint32_t b) {
    if (b < 26) {
        if (b < 5) return (uint32_t)("@SYN."[b]);
        if (b < 17) {
            unsigned long long v = rec % 1000000000000ull;
            for (uint32_t k = 16; k > b; --k) v /= 10;
            return '0' + (uint32_t)(v % 10);
        }
        return (uint32_t)(" 1:N:0:1\n"[b - 17]);
    }
    if (b < 176) {
        unsigned long long h = synth_hash(seed, rec, 1, b - 26);
        if ((uint32_t)(h >> 32) % 100u == 0) return 'N';
        return (uint32_t)("ACGT"[h & 3]);
    }
    if (b == 176) return '\n';
    if (b == 177) return '+';
    if (b == 178) return '\n';
    if (b < 329) {
        unsigned long long h = synth_hash(seed, rec, 2, b - 179);
        return '#' + (uint32_t)(h >> 32) % 39u;
    }
    return '\n';
}
__global__ __launch_bounds__(256) void k_synth(uint8_t *__restrict__ out, uint64_t byte_off, uint64_t len,
                                               unsigned long long seed) {
    const uint64_t nchunks = (len + 15) / 16;
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < nchunks;
         c += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t o = c * 16;
        unsigned long long pos = byte_off + o;
        unsigned long long rec = pos / 330;
        uint32_t b = (uint32_t)(pos % 330);
        uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
        const uint32_t n = (len - o) < 16 ? (uint32_t)(len - o) : 16u;
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t v = synth_byte(seed, rec, b) << ((i & 3) * 8);
            if (i < 4) w0 |= v; else if (i < 8) w1 |= v; else if (i < 12) w2 |= v; else w3 |= v;
            if (++b == 330) { b = 0; ++rec; }
        }
        if (n == 16) {
            *reinterpret_cast<uint4 *>(out + o) = make_uint4(w0, w1, w2, w3);
        } else {
            for (uint32_t i = 0; i < n; ++i) {
                const uint32_t w = i < 4 ? w0 : i < 8 ? w1 : i < 12 ? w2 : w3;
                out[o + i] = (uint8_t)(w >> ((i & 3) * 8));
            }
        }
    }
}
void launch_synth(hipStream_t s, uint8_t *out, uint64_t byte_off, uint64_t len, uint64_t seed) {
    if (!len) return;
    uint64_t blocks = ((len + 15) / 16 + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_synth, dim3((uint32_t)blocks), dim3(256), 0, s, out, byte_off, len,
                       (unsigned long long)seed);
}

// ---------------------------------------------------------------------------------------------
// Streaming-read ceiling: the same access pattern as k_index (a wavefront reads a 16 KiB tile as
// 1 KiB pieces, four 16-byte loads in flight per lane) with only an integer sum as work.  Persistent
// grid so that the final atomics do not serialise.
__global__ __launch_bounds__(256) void k_read_ceiling(const uint8_t *__restrict__ buf, uint64_t len,
                                                      unsigned long long *__restrict__ sum) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t n_tiles = (len + WT_BYTES - 1) >> WT_SHIFT;
    const uint64_t wave0 = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t nwaves = (uint64_t)gridDim.x * 4;
    unsigned long long acc = 0;
    for (uint64_t tile = wave0; tile < n_tiles; tile += nwaves) {
        const uint64_t tbase = tile << WT_SHIFT;
        if (tbase + WT_BYTES <= len) {
#pragma unroll 1
            for (uint32_t g = 0; g < WT_PIECES / 4; ++g) {
                const uint8_t *p = buf + tbase + g * 4 * PIECE_BYTES + lane * 16;
                typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 v0 = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
                const u32x4 v1 = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p + PIECE_BYTES));
                const u32x4 v2 = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p + 2 * PIECE_BYTES));
                const u32x4 v3 = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p + 3 * PIECE_BYTES));
                acc += (unsigned long long)v0.x + v0.y + v0.z + v0.w;
                acc += (unsigned long long)v1.x + v1.y + v1.z + v1.w;
                acc += (unsigned long long)v2.x + v2.y + v2.z + v2.w;
                acc += (unsigned long long)v3.x + v3.y + v3.z + v3.w;
            }
        } else {
            for (uint64_t o = tbase + lane * 4; o + 4 <= len && o < tbase + WT_BYTES; o += 256)
                acc += *reinterpret_cast<const uint32_t *>(buf + o);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    if (lane == 0 && acc) atomicAdd(sum, acc);
}
void launch_read_ceiling(hipStream_t s, const uint8_t *buf, uint64_t len, uint64_t *sum, int n_cu) {
    const uint64_t n_tiles = (len + WT_BYTES - 1) / WT_BYTES;
    if (!n_tiles) return;
    uint64_t blocks = (n_tiles + 3) / 4;
    const uint64_t maxb = (uint64_t)(n_cu > 0 ? n_cu : 256) * 8;
    if (blocks > maxb) blocks = maxb;
    hipLaunchKernelGGL(k_read_ceiling, dim3((uint32_t)blocks), dim3(256), 0, s, buf, len,
                       (unsigned long long *)sum);
}

}  // namespace fqh
