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
