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
// k_stats_lines — the second histogram kernel (one line per lane); superseded by k_stats_oct below and kept
// as FQH_STATS_VARIANT=1, an independent statement of the same result for cross-checks.
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
constexpr uint32_t SO_LISTW = 512;            // list entries staged in LDS per wave (u16 each); 256 when extra rows need the room
constexpr uint32_t SO_LX_MAX = 256;           // extra rows (columns 256 .. 511) in a plain [row][72] layout, exact path only
constexpr uint32_t SO_LDS_MAX = 160 * 1024;
constexpr uint32_t SO_ADDR_SPAN = 65536 + SO_SBYTES + 128 + 3 * 16384;  // see launch_stats_oct

__device__ __forceinline__ uint32_t so_slot(uint32_t r) {  // r = row % 64
    return ((r >> 2) & 7u) | ((r & 3u) << 3) | (r & 32u);
}
__device__ __forceinline__ uint32_t so_row6(uint32_t slot) {
    return ((slot & 7u) << 2) | ((slot >> 3) & 3u) | (slot & 32u);
}
// word index of (bin, row): quality bins 0..63, sequence bins 0..7
template <bool IS_SEQ>
__device__ __forceinline__ uint32_t so_word(uint32_t bin, uint32_t row) {
    const uint32_t rb = row >> 6, slot = so_slot(row & 63u);
    return IS_SEQ ? ((rb << 9) | (bin << 6) | slot) : SO_SBYTES / 4 + ((rb << 12) | (bin << 6) | slot);
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

// Columns a.lc .. a.lc + a.lx - 1 (reads longer than the 256 bank-scheduled rows) have plain LDS rows of
// 72 words (64 quality bins, 8 sequence bins) behind the staged lists; only the exact path touches them.
__device__ __forceinline__ uint32_t *so_extra(const StatsArgs &a, uint32_t *hist) {
    return hist + SO_WORDS + (SO_WAVES * a.listw) / 2;
}

// The exact per-byte statement: columns pos .. of a line of `len` columns held in w.  (lc is the tile's
// view of the bank-scheduled rows: 0 in tiles that take the exact path for everything.)
template <bool IS_SEQ>
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
            if (col < lc) atomicAdd(hist + so_word<true>(bin, col), 1u);
            else if (col - a.lc < a.lx) atomicAdd(so_extra(a, hist) + (col - a.lc) * 72u + 64u + bin, 1u);
            else if (col < a.lmax) atomicAdd(&a.base_hist[(uint64_t)col * 8 + bin_to_class(bin)], 1ull);
            else atomicAdd(&a.scalars[IS_SEQ ? 5 : 6], 1ull);  // a column beyond the caller's lmax
        } else {
            if (col < lc && b - 33u < 64u) atomicAdd(hist + so_word<false>(b - 33u, col), 1u);
            else if (col - a.lc < a.lx && b - 33u < 64u) atomicAdd(so_extra(a, hist) + (col - a.lc) * 72u + (b - 33u), 1u);
            else if (col < a.lmax) atomicAdd(&a.qual_hist[(uint64_t)col * 256 + b], 1ull);
            else atomicAdd(&a.scalars[IS_SEQ ? 5 : 6], 1ull);  // a column beyond the caller's lmax
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
};
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
    (void)__hip_atomic_fetch_sub((so_lds_u32 *)(byte_addr + OFF), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Count one batch (its loads were issued one batch earlier), straight-line: pass 1 checks every byte
// the batch counts (whole dwords under the lane's masks, the partial tail over filler bytes), pass 2
// adds them -- one v_perm_b32 and one ds_sub per byte; lanes without a whole dword subtract 0 at
// whatever address their bytes give (the kernel's LDS allocation covers every address a byte can
// form).  A byte outside the window / alphabet sends the whole batch to the exact path instead.
template <bool IS_SEQ, uint32_t NSL, bool DBG>
__device__ __forceinline__ void so_count(const StatsArgs &a, const uint8_t *tbase, SoBatch<NSL> &B, SoShape<NSL> &S,
                                         uint32_t lane, uint32_t lc, uint32_t *hist, const SoLane &c, uint32_t my_len,
                                         uint32_t src4, SoTotals &T, SoAcc &acc, bool trimmed, bool &cr_seen) {
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
            T.not_dna += so_groups(bn);
            T.not_dnan += so_groups(bi);
        }
    }
}

template <uint32_t NSL, bool DBG>
__global__ __launch_bounds__(SO_THREADS) void k_stats_oct(StatsArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];  // sequence + quality regions, the staged lists, the extra rows
    const uint32_t lc = a.lc;
    const uint32_t listw = a.listw;  // 512, or 256 when the extra rows need the room
    for (uint32_t i = threadIdx.x; i < SO_WORDS; i += SO_THREADS) hist[i] = 0;
    for (uint32_t i = threadIdx.x; i < a.lx * 72u; i += SO_THREADS) so_extra(a, hist)[i] = 0;
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // a scalar: so is the tile
    const uint32_t m4 = (lane & 7u) * 4u, g8 = lane >> 3;
    uint16_t *const wl = reinterpret_cast<uint16_t *>(hist + SO_WORDS) + wv * listw;  // this wave's staged list
    // The address registers assume the histogram starts at LDS address 0 (it is the kernel's only
    // LDS object); a shared-memory pointer is its LDS address in the low 32 bits.
    if ((uint32_t)(uintptr_t)hist != 0) __builtin_trap();
    SoLane c;
    c.slots = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t j = k ^ (g8 & 3u);
        c.sel[k] = 0x0C0C0004u + k + (j << 8);
        c.slots |= (((lane & 7u) + 8u * j) * 4u) << (8u * k);
    }
    SoAcc acc = {0, 0, 0};
    SoTotals T = {0, 0};
    bool cr_seen = false;  // wave-uniform: a line of this wave's tiles ended in "\r\n" -- look for it from the next tile on
    // DBG (FQH_STATS_DBG & 8192): cycles this wave spent waiting for a tile's words, staging its list, working out
    // its lines, and in its batches (added to qual_hist[0..3] at the end; tools/exp_statsdbg.py prints them)
    unsigned long long dbgt[4] = {0, 0, 0, 0};
    SoShape<NSL> S = {};
    S.key = 0xFFFFFFFFu;  // no P has this key: the first batch works the shape out

    // What a wave needs to know about a tile before it can start on it, loaded one tile ahead (the
    // per-tile chain count -> prefix -> list -> '\r' bytes -> first dwords is otherwise paid in full,
    // 256 times per wave: 2.5 of the kernel's 6 ms).
    struct TilePre {
        uint32_t meta;  // lane 0: the tile's count, 1: its prefix, 2: the next tile's count, 3: that tile's first
                        // entry, 4-5: the block prefix -- six words, one load, one register
        uint2 l0, l1;   // list entries 4 lane .. 4 lane + 3 and 256 + 4 lane .. + 3
    };
    auto prefetch = [&](uint64_t t, TilePre &P) {
        const uint64_t tc = t < a.n_tiles ? t : a.n_tiles - 1;  // clamped: the loads are unconditional
        const uint16_t *__restrict__ tl = a.list + tc * a.list_cap;
        P.l0 = *reinterpret_cast<const uint2 *>(tl + lane * 4);
        P.l1 = *reinterpret_cast<const uint2 *>(tl + 256 + lane * 4);  // list_cap >= 512
        const uint64_t t1 = tc + 1 < a.n_tiles ? tc + 1 : tc;
        const uint32_t *mp = a.tile_count + tc;
        if (lane == 1) mp = a.tile_prefix + tc;
        if (lane == 2) mp = a.tile_count + t1;
        if (lane == 3) mp = reinterpret_cast<const uint32_t *>(a.list + t1 * a.list_cap);  // (list_cap is even)
        if (lane == 4 || lane == 5)
            mp = reinterpret_cast<const uint32_t *>(a.block_prefix + (tc >> SCAN_SHIFT)) + (lane - 4);
        P.meta = *mp;
    };
    const uint64_t tstride = (uint64_t)gridDim.x * SO_WAVES;
    uint64_t tile = (uint64_t)blockIdx.x * SO_WAVES + wv;
    TilePre nextP;
    if (tile < a.n_tiles && a.len >= 4) prefetch(tile, nextP);
    if (DBG && (a.dbg & 65536u) && wv >= 8) tile = a.n_tiles;  // half the waves idle: does the other half get faster?
    for (; tile < a.n_tiles && a.len >= 4; tile += tstride) {
        const TilePre cur = nextP;
        unsigned long long tk0 = 0, tk1 = 0, tk2 = 0, tk3 = 0;
        if (DBG) tk0 = __builtin_readcyclecounter();
        prefetch(tile + tstride, nextP);
        // (six lanes hold the tile's six words: as scalars, the tile's bookkeeping runs on the scalar unit)
        uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)cur.meta, 0);
        cnt = cnt < a.list_cap ? cnt : a.list_cap;
        if (cnt == 0) continue;
        const uint32_t cur_tp = (uint32_t)__builtin_amdgcn_readlane((int)cur.meta, 1);
        const uint32_t cur_cnt1 = (uint32_t)__builtin_amdgcn_readlane((int)cur.meta, 2);
        const uint32_t cur_first1 = (uint32_t)__builtin_amdgcn_readlane((int)cur.meta, 3);
        const unsigned long long cur_bp =
            (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)cur.meta, 4) |
            ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)cur.meta, 5) << 32);
        const unsigned long long lbase = a.nl_count + 1 + cur_bp + cur_tp;
        if (lbase >= a.line_hi || lbase + cnt <= a.line_lo) continue;
        if (DBG && (a.dbg & 1024u)) { acc.rec += cnt; continue; }  // only the walk over the tiles' words
        if (DBG) { tk1 = __builtin_readcyclecounter(); dbgt[0] += tk1 - tk0; }
        const uint16_t *__restrict__ tl = a.list + tile * a.list_cap;
        const uint64_t tb = tile << WT_SHIFT;
        // stage the list
        __builtin_amdgcn_wave_barrier();
        *reinterpret_cast<uint2 *>(wl + lane * 4) = cur.l0;
        if (listw > 256) *reinterpret_cast<uint2 *>(wl + 256 + lane * 4) = cur.l1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (DBG) { tk2 = __builtin_readcyclecounter(); dbgt[1] += tk2 - tk1; }
        // start of the first line after this tile (ends the tile's last line), tile-relative
        uint64_t next_first = a.valid_end;
        if (tile + 1 < a.n_tiles) {
            if (cur_cnt1) {
                next_first = ((tile + 1) << WT_SHIFT) + (cur_first1 & 0x3FFFu);
            } else {
                for (uint64_t t2 = tile + 2; t2 < a.n_tiles; ++t2) {
                    if (a.tile_count[t2]) { next_first = (t2 << WT_SHIFT) + (a.list[t2 * a.list_cap] & 0x3FFFu); break; }
                }
            }
        }
        const uint64_t vend = a.valid_end > tb ? a.valid_end - tb : 0;
        const uint32_t vend_rel = vend < 0x7FFFFFFFull ? (uint32_t)vend : 0x7FFFFFFFu;
        const uint64_t nf = next_first > tb ? next_first - tb : 0;
        const uint32_t nf_rel = nf < 0x7FFFFFFFull ? (uint32_t)nf : 0x7FFFFFFFu;
        // entries of this tile whose lines count: global line index in [line_lo, line_hi)
        const uint32_t e_lo = a.line_lo > lbase ? (uint32_t)(a.line_lo - lbase) : 0u;
        const uint32_t e_hi = a.line_hi - lbase < cnt ? (uint32_t)(a.line_hi - lbase) : cnt;
        const uint32_t lb3 = (uint32_t)lbase & 3u;
        const uint8_t *const tbase = a.buf + tb;
        // every unconditional load of a line that starts in this tile stays inside the buffer; the
        // (at most two) tiles at the end of the buffer for which that does not hold take the exact
        // path for everything
        const bool safe = tb + WT_BYTES + 32u * NSL + 8u <= a.len;
        const uint32_t lce = safe ? lc : 0u;  // LDS rows in use for this tile: none => every column is exact

        // one lane per line: start and raw length of line `sbl` of `kind`; false: no line that counts
        auto line_of = [&](uint32_t kind, uint32_t sbl, uint32_t &s_rel, uint32_t &len) -> bool {
            const uint32_t i0 = ((kind ? 3u : 1u) - lb3) & 3u;
            const uint32_t i = i0 + 4u * sbl;
            if (i >= cnt || i < e_lo || i >= e_hi) return false;
            s_rel = (i < listw ? wl[i] : tl[i]) & 0x3FFFu;
            uint32_t n_rel = i + 1 < cnt ? ((i + 1 < listw ? wl[i + 1] : tl[i + 1]) & 0x3FFFu) : nf_rel;
            n_rel = n_rel < vend_rel ? n_rel : vend_rel;
            len = n_rel - 1 - s_rel;  // raw line, without its '\n'
            return true;
        };
        const uint32_t i0s = (1u - lb3) & 3u, i0q = (3u - lb3) & 3u;
        const uint32_t nls = i0s < cnt ? (cnt - i0s + 3) >> 2 : 0u, nlq = i0q < cnt ? (cnt - i0q + 3) >> 2 : 0u;

        if (safe && nls <= 64 && nlq <= 64 && !(DBG && (a.dbg & 8u))) {
            // ---- the usual tile: at most 64 lines of each kind.  Both kinds' lines are worked out at
            // once (their '\r' bytes are in flight together), then all batches run as one sequence so that
            // the first quality batch is fetched while the last sequence batch is counted.
            uint32_t s_s = 0, l_s = 0, s_q = 0, l_q = 0;
            const bool has_s = line_of(0, lane, s_s, l_s), has_q = line_of(1, lane, s_q, l_q);
            // The byte before each line's '\n' is only looked at (two scattered loads per tile, and their latency before the
            // first batch can be packed) once the wave has met a "\r\n"; until then so_count's exact path does the trimming.
            const bool probe = cr_seen && !(DBG && (a.dbg & 16u));
            const uint32_t cr_s = (probe && has_s && l_s) ? tbase[s_s + l_s - 1] : 0u;
            const uint32_t cr_q = (probe && has_q && l_q) ? tbase[s_q + l_q - 1] : 0u;
            if (cr_s == '\r') --l_s;                                                   // trim_winline, src/records.rs:66-73
            if (cr_q == '\r') --l_q;
            const uint32_t P_s = has_s ? so_pack(s_s, l_s, lce) : 0u, P_q = has_q ? so_pack(s_q, l_q, lce) : 0u;
            if (has_s) { ++acc.rec; acc.bases += l_s; }
            if (has_q) acc.qual += l_q;
            if (DBG) { tk3 = __builtin_readcyclecounter(); dbgt[2] += tk3 - tk2; }
            const uint32_t nbs = (nls + 7) >> 3, nbq = (DBG && (a.dbg & 64u)) ? 0u : (nlq + 7) >> 3;  // 64: sequence lines only
            // Batch f = 2 k + kind: the k-th eight sequence lines, then the k-th eight quality lines -- the lines of (nearly) the
            // same records, a few hundred bytes apart, so the second batch's loads find the first one's cache lines.  A kind
            // that has run out of lines gets an empty batch (P = 0 counts nothing).
            const uint32_t nbm = nbs > nbq ? nbs : nbq;
            const uint32_t nbt = (DBG && (a.dbg & 512u)) ? 0u : 2u * nbm;                               // 512: no batches
            // The line words of a batch are looked up a batch early, before the atomics of the batch being counted:
            // the ds_bpermute then does not queue behind them, and the loads go out as soon as their turn comes.
            auto lookup = [&](uint32_t f) -> uint32_t {
                const bool isq = (f & 1u) != 0;
                const uint32_t b = f >> 1;
                uint32_t P = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(32u * b + 4u * g8), (int)(isq ? P_q : P_s));
                if (DBG && (a.dbg & 32768u))  // every group reads the first group's line: an eighth of the cache lines
                    P = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(32u * b), (int)(isq ? P_q : P_s));
                if (b >= (isq ? nbq : nbs)) P = 0;
                __builtin_amdgcn_sched_barrier(0);  // (keep it where it is: ahead of the count's atomics)
                return P;
            };
            auto fetch = [&](uint32_t P, SoBatch<NSL> &B) {
                B.P = P;
                const uint32_t o = (B.P >> SO_P_SREL) + m4;
                if (DBG && (a.dbg & 384u)) {  // 128: only the first step's load, 256: none
#pragma unroll
                    for (uint32_t u = 0; u < NSL; ++u) B.w[u] = o;
                    if (a.dbg & 128u) B.w[0] = load4_fast(tbase + o);
                    return;
                }
#pragma unroll
                for (uint32_t u = 0; u < NSL; ++u) B.w[u] = load4_fast(tbase + (o + 32 * u));
                if (DBG && (a.dbg & 16384u)) {  // the same loads once more, one byte on: what does a load that hits cost?
#pragma unroll
                    for (uint32_t u = 0; u < NSL; ++u) B.w[u] ^= load4_fast(tbase + (o + 32 * u + 1));
                }
            };
            auto count_s = [&](uint32_t f, SoBatch<NSL> &B) {  // f even
                if (DBG && (a.dbg & 4u)) { acc.rec += B.w[0] == 0x12345u; return; }
                so_count<true, NSL, DBG>(a, tbase, B, S, lane, lce, hist, c, l_s, 32u * (f >> 1) + 4u * g8, T, acc, probe, cr_seen);
            };
            auto count_q = [&](uint32_t f, SoBatch<NSL> &B) {  // f odd
                if (DBG && (a.dbg & 4u)) { acc.rec += B.w[0] == 0x12345u; return; }
                so_count<false, NSL, DBG>(a, tbase, B, S, lane, lce, hist, c, l_q, 32u * (f >> 1) + 4u * g8, T, acc, probe, cr_seen);
            };
            // The fetches are unconditional inside the loops (the index is clamped instead) so that the
            // compiler's s_waitcnt for the batch it needs leaves the next one's loads in flight.
            if (nbt) {
                SoBatch<NSL> B0, B1;  // ping-pong: the loads of one are in flight while the other is counted
                const uint32_t fl = nbt - 1;  // (nbt is even: B0 holds the sequence batches, B1 the quality batches)
                uint32_t pa = lookup(0), pb = lookup(1);  // the words of the next even / odd batch
                fetch(pa, B0);
                for (uint32_t f = 0; f < nbt; f += 2) {
                    fetch(pb, B1);
                    pa = lookup(f + 2 < fl ? f + 2 : fl);  // (past the end: the last batch once more, not counted)
                    count_s(f, B0);
                    fetch(pa, B0);
                    pb = lookup(f + 3 < fl ? f + 3 : fl);
                    count_q(f + 1, B1);
                }
            }
            if (DBG) dbgt[3] += __builtin_readcyclecounter() - tk3;
            continue;
        }

        // ---- any other tile (more than 64 lines of a kind, or too close to the end of the buffer for
        // unconditional loads): one kind after the other, 64 lines at a time
        for (uint32_t kind = 0; kind < 2; ++kind) {        // 0: sequence lines, 1: quality lines
            const uint32_t nlines = kind ? nlq : nls;
            if (!nlines) continue;
            for (uint32_t sb = 0; sb < nlines; sb += 64) {
                uint32_t my_P = 0, my_len = 0;
                {
                    uint32_t s_rel = 0, len = 0;
                    if (sb + lane < nlines && line_of(kind, sb + lane, s_rel, len)) {
                        if (len && tbase[s_rel + len - 1] == '\r') --len;              // trim_winline, src/records.rs:66-73
                        my_len = len;
                        my_P = so_pack(s_rel, len, lce);
                        if (kind == 0) {
                            ++acc.rec;
                            acc.bases += len;
                        } else {
                            acc.qual += len;
                        }
                    }
                }
                const uint32_t nbat = ((nlines - sb < 64 ? nlines - sb : 64u) + 7) >> 3;
                SoBatch<NSL> B0;
                for (uint32_t b = 0; b < nbat; ++b) {
                    B0.P = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(32u * b + 4u * g8), (int)my_P);
                    const uint32_t o = (B0.P >> SO_P_SREL) + m4;
                    if (safe) {
#pragma unroll
                        for (uint32_t u = 0; u < NSL; ++u) B0.w[u] = load4_fast(tbase + (o + 32 * u));
                    } else {  // lce == 0: every column goes through the exact path, which loads for itself
#pragma unroll
                        for (uint32_t u = 0; u < NSL; ++u) B0.w[u] = 0;
                    }
                    if (kind == 0) so_count<true, NSL, DBG>(a, tbase, B0, S, lane, lce, hist, c, my_len, 32u * b + 4u * g8, T, acc, true, cr_seen);
                    else so_count<false, NSL, DBG>(a, tbase, B0, S, lane, lce, hist, c, my_len, 32u * b + 4u * g8, T, acc, true, cr_seen);
                }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the inline ds_add of lds_add
    __syncthreads();
    uint32_t *__restrict__ dst = a.scratch + (uint64_t)blockIdx.x * (SO_WORDS + a.lx * 72u);
    for (uint32_t i = threadIdx.x; i < SO_WORDS; i += SO_THREADS) dst[i] = hist[i];
    for (uint32_t i = threadIdx.x; i < a.lx * 72u; i += SO_THREADS) dst[SO_WORDS + i] = so_extra(a, hist)[i];
    if (DBG && (a.dbg & 8192u) && lane == 0)
        for (int j = 0; j < 4; ++j) atomicAdd(&a.qual_hist[j], dbgt[j]);
    // per-line totals: rec / bases / qual were summed by the lane that owned the line; the two
    // "not DNA" counts are wave-uniform
    unsigned long long sc[5] = {acc.rec, acc.bases, acc.qual, 0, 0};
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
            if (sc[j]) atomicAdd(&a.scalars[j], sc[j]);
    }
}

// Sum the per-block partial histograms into the caller's u64 arrays (coalesced reads).
__global__ __launch_bounds__(256) void k_stats_reduce_oct(const uint32_t *__restrict__ scratch, uint32_t n_blocks,
                                                          uint32_t lc, uint32_t lx,
                                                          unsigned long long *__restrict__ qual_hist,
                                                          unsigned long long *__restrict__ base_hist) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t words = SO_WORDS + lx * 72u;
    if (id >= words) return;
    bool isq;
    uint32_t bin, row;
    if (id < SO_WORDS) {  // bank-scheduled rows
        isq = id >= SO_SBYTES / 4;
        const uint32_t r = isq ? id - SO_SBYTES / 4 : id;
        const uint32_t rb = isq ? r >> 12 : r >> 9;
        bin = isq ? (r >> 6) & 63u : (r >> 6) & 7u;
        row = rb * 64 + so_row6(r & 63u);
        if (row >= lc) return;
    } else {              // plain rows lc .. lc + lx - 1: 64 quality bins, 8 sequence bins
        const uint32_t r = id - SO_WORDS;
        row = lc + r / 72u;
        isq = r % 72u < 64u;
        bin = isq ? r % 72u : r % 72u - 64u;
    }
    const uint32_t b0 = blockIdx.y * RED_GROUP;
    const uint32_t b1 = b0 + RED_GROUP < n_blocks ? b0 + RED_GROUP : n_blocks;
    unsigned long long s = 0;
    for (uint32_t b = b0; b < b1; ++b) s += scratch[(uint64_t)b * words + id];
    if (!s) return;
    if (isq) atomicAdd(&qual_hist[(uint64_t)row * 256 + 33 + bin], s);
    else atomicAdd(&base_hist[(uint64_t)row * 8 + bin_to_class(bin)], s);  // bins 0,2,5 share class 5
}

uint32_t stats_oct_lc(uint32_t lmax) { return lmax < SO_LC_MAX ? lmax : SO_LC_MAX; }
static uint32_t stats_oct_lx(uint32_t lmax) {
    return lmax > SO_LC_MAX ? (lmax - SO_LC_MAX < SO_LX_MAX ? lmax - SO_LC_MAX : SO_LX_MAX) : 0u;
}
size_t stats_oct_scratch_bytes(uint32_t lmax, int n_cu) {
    return (size_t)stats_lines_blocks(n_cu) * (SO_WORDS + stats_oct_lx(lmax) * 72u) * sizeof(uint32_t);
}
template <uint32_t NSL, bool DBG>
static hipError_t launch_stats_oct_n(hipStream_t s, const StatsArgs &a, uint32_t blocks, size_t lds) {
    // Lanes without a whole dword subtract 0 at the address their bytes happen to form: any bin byte
    // (< 64 KiB) plus the largest row-block offset.  The allocation covers all of them.
    if (lds < SO_ADDR_SPAN) lds = SO_ADDR_SPAN;
    if (lds > SO_LDS_MAX) return hipErrorInvalidValue;
    static size_t set = 0;
    if (lds > set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_stats_oct<NSL, DBG>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        set = lds;
    }
    hipLaunchKernelGGL((k_stats_oct<NSL, DBG>), dim3(blocks), dim3(SO_THREADS), lds, s, a);
    return hipSuccess;
}
hipError_t launch_stats_oct(hipStream_t s, StatsArgs a, int n_cu) {
    a.lc = stats_oct_lc(a.lmax);
    static const uint32_t dbg = getenv("FQH_STATS_DBG") ? (uint32_t)atoi(getenv("FQH_STATS_DBG")) : 0u;
    a.dbg = dbg;
    a.lx = stats_oct_lx(a.lmax);
    a.listw = a.lx ? 256u : SO_LISTW;
    size_t lds = (size_t)SO_WORDS * sizeof(uint32_t) + (size_t)SO_WAVES * a.listw * sizeof(uint16_t) +
                 (size_t)a.lx * 72 * sizeof(uint32_t);
    if (lds > SO_LDS_MAX) {  // cannot happen with the constants above; keep the kernel launchable anyway
        a.lx = 0;
        lds = (size_t)SO_WORDS * sizeof(uint32_t) + (size_t)SO_WAVES * a.listw * sizeof(uint16_t);
    }
    const uint32_t blocks = stats_lines_blocks(n_cu);
    const uint32_t nsl = (a.lc + 31) / 32;  // steps that hold LDS rows
    hipError_t e = dbg        ? launch_stats_oct_n<5, true>(s, a, blocks, lds)   // timing experiments: 150-bp shape only
                   : nsl <= 2 ? launch_stats_oct_n<2, false>(s, a, blocks, lds)
                   : nsl <= 4 ? launch_stats_oct_n<4, false>(s, a, blocks, lds)
                   : nsl <= 5 ? launch_stats_oct_n<5, false>(s, a, blocks, lds)
                              : launch_stats_oct_n<8, false>(s, a, blocks, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_stats_reduce_oct, dim3((SO_WORDS + a.lx * 72u + 255) / 256, (blocks + RED_GROUP - 1) / RED_GROUP),
                       dim3(256), 0, s, a.scratch, blocks, a.lc, a.lx, a.qual_hist, a.base_hist);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// k_stats_head — the record in progress at the chunk start, for callers whose buffer also holds its
// beginning in front of the chunk (fqh_stats_launch_lead; the streaming ring).  One wavefront.  The
// record's line starts before the chunk come from the carry (distances back[]), those inside from
// the full line lists; it ends at the line start with global index 4 (r0 + 1), or at the end of the
// chunk.  Counting is the plain per-byte statement on the caller's u64 arrays.
__global__ __launch_bounds__(64) void k_stats_head(StatsArgs a, unsigned long long b0, unsigned long long b1,
                                                   unsigned long long b2, unsigned long long b3) {
    const uint32_t lane = threadIdx.x;
    const unsigned long long back[4] = {b0, b1, b2, b3};
    const unsigned long long r0 = a.nl_count >> 2;
    auto entry = [&](unsigned long long j, long long &off) -> bool {  // j-th line start of the chunk
        unsigned long long cum = 0;
        for (uint64_t t = 0; t < a.n_tiles; ++t) {
            uint32_t c = a.tile_count[t];
            c = c < a.list_cap ? c : a.list_cap;
            if (j < cum + c) {
                off = (long long)((t << WT_SHIFT) + (a.list[t * a.list_cap + (uint32_t)(j - cum)] & 0x3FFFu));
                return true;
            }
            cum += c;
        }
        return false;
    };
    long long p[5];
    bool ok = true;
    for (int i = 0; i < 5; ++i) {
        const unsigned long long g = 4 * r0 + i;  // global line index
        if (g <= a.nl_count) {
            p[i] = -(long long)back[a.nl_count - g];
        } else if (!entry(g - a.nl_count - 1, p[i])) {
            if (i == 4) p[i] = (long long)a.len;  // the record's last '\n' is the last byte of the chunk
            else ok = false;
        }
    }
    if (!ok) return;
    const uint8_t *const base = a.buf;
    unsigned long long n_bases = 0, n_qual = 0, oseq = 0, oqual = 0;
    uint32_t any_n = 0, any_inv = 0;
    for (int kind = 0; kind < 2; ++kind) {
        const long long s = kind ? p[3] : p[1];
        long long len = (kind ? p[4] : p[2]) - 1 - s;            // raw line, without its '\n'
        if (len > 0 && base[s + len - 1] == '\r') --len;         // trim_winline, src/records.rs:66-73
        if (len < 0) len = 0;
        if (kind) n_qual = (unsigned long long)len; else n_bases = (unsigned long long)len;
        for (long long col = lane; col < len; col += 64) {
            const uint32_t b = base[s + col];
            if (kind == 0) {
                const uint32_t c = base_class(b);
                any_inv |= c == 5 ? 1u : 0u;
                any_n |= c == 4 ? 1u : 0u;
                if (col < (long long)a.lmax) atomicAdd(&a.base_hist[(uint64_t)col * 8 + c], 1ull);
                else ++oseq;
            } else {
                if (col < (long long)a.lmax) atomicAdd(&a.qual_hist[(uint64_t)col * 256 + b], 1ull);
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
        atomicAdd(&a.scalars[0], 1ull);
        if (n_bases) atomicAdd(&a.scalars[1], n_bases);
        if (n_qual) atomicAdd(&a.scalars[2], n_qual);
        if (!gi && !gn) atomicAdd(&a.scalars[3], 1ull);
        if (!gi) atomicAdd(&a.scalars[4], 1ull);
        if (oseq) atomicAdd(&a.scalars[5], oseq);
        if (oqual) atomicAdd(&a.scalars[6], oqual);
    }
}
void launch_stats_head(hipStream_t s, const StatsArgs &a, const uint64_t back[4]) {
    hipLaunchKernelGGL(k_stats_head, dim3(1), dim3(64), 0, s, a, (unsigned long long)back[0],
                       (unsigned long long)back[1], (unsigned long long)back[2], (unsigned long long)back[3]);
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
