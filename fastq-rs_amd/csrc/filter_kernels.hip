// filter_kernels.hip — scan -> select -> gather -> write: per-record alphabet flags
// (Record::validate_dna / validate_dnan, src/records.rs:19-33) and the compaction of the selected
// records into one contiguous output, i.e. what a filter built on Record::write (src/records.rs:93-96,
// RefRecord::write copies the record's raw bytes) produces on the CPU.
//
// Input is the IdxRecord-style index of fqh_index_records (start + the four newline offsets).
#include <hip/hip_runtime.h>

#include "fqh_internal.h"

namespace fqh {

// ---------------------------------------------------------------------------------------------
// k_record_flags: 8 lanes per record, 8 records per wavefront.  Lane m checks dwords m, m + 8, ... of
// the sequence line with the same SWAR test as the histogram kernel (v_perm_b32 as an 8-entry LUT:
// a byte is in ACGTN iff it equals LUT[byte & 7]); bit 0 = all of ACGT, bit 1 = all of ACGTN.
__global__ __launch_bounds__(256) void k_record_flags(const uint8_t *__restrict__ buf, uint64_t len,
                                                      uint64_t base_offset, const fqh_idx_record *__restrict__ idx,
                                                      uint64_t n, uint8_t *__restrict__ flags) {
    const uint32_t lane = threadIdx.x & 63u, m = lane & 7u, g = lane >> 3;
    const uint64_t rec = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + g;
    uint32_t bad = 0, has_n = 0;  // per lane: a byte outside ACGTN / an 'N'
    if (rec < n) {
        const fqh_idx_record r = idx[rec];
        const uint64_t s = r.start - base_offset + r.head + 1;   // first sequence byte, chunk-relative
        uint32_t sl = r.seq - r.head - 1;
        if (sl && buf[s + sl - 1] == '\r') --sl;                   // trim_winline, src/records.rs:66-73
        const uint8_t *p = buf + s;
        for (uint32_t c = m * 4; c < sl; c += 32) {
            uint32_t w = 0;
            const uint32_t nb = sl - c < 4 ? sl - c : 4u;
            if (s + c + 4 <= len) {
                __builtin_memcpy(&w, p + c, 4);
            } else {
                for (uint32_t i = 0; i < nb; ++i) w |= (uint32_t)p[c + i] << (8 * i);
            }
            if (nb < 4) {  // bytes past the line's end become 'A'
                const uint32_t keep = (1u << (8 * nb)) - 1u;
                w = (w & keep) | (0x41414141u & ~keep);
            }
            const uint32_t bins = w & 0x07070707u;
            bad |= w ^ __builtin_amdgcn_perm(0x474EFF54u, 0x43FF41FFu, bins);
            has_n |= __builtin_amdgcn_perm(0x00800000u, 0u, bins);  // bin 6 = 'N' (valid bytes only matter)
        }
    }
    const unsigned long long bb = __ballot(bad != 0), bn = __ballot(has_n != 0);
    if (rec < n && m == 0) {
        const bool inv = ((bb >> (lane & 56u)) & 0xFFu) != 0, nn = ((bn >> (lane & 56u)) & 0xFFu) != 0;
        flags[rec] = (uint8_t)((!inv && !nn ? 1u : 0u) | (!inv ? 2u : 0u));
    }
}

// ---------------------------------------------------------------------------------------------
// Gather.  Records are taken in blocks of GB_RECS; pass 1 sums the selected records' sizes per block,
// pass 2 (one block) scans the block sums, pass 3 recomputes the in-block prefix and copies.
constexpr uint32_t GB_RECS = 1024;  // records per block: 4 per thread

__device__ __forceinline__ uint32_t sel_size(const fqh_idx_record &r, uint8_t f, uint32_t mask, uint32_t want) {
    return (f & mask) == want ? r.qual + 1u : 0u;  // the record's raw bytes including its last '\n'
}

__global__ __launch_bounds__(256) void k_gather_sizes(const fqh_idx_record *__restrict__ idx, uint64_t n,
                                                      const uint8_t *__restrict__ flags, uint32_t mask, uint32_t want,
                                                      unsigned long long *__restrict__ block_bytes,
                                                      unsigned long long *__restrict__ block_recs) {
    __shared__ unsigned long long sb[4], sr[4];
    const uint64_t r0 = (uint64_t)blockIdx.x * GB_RECS + threadIdx.x * 4;
    unsigned long long bytes = 0, recs = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (r0 + i < n) {
            const uint32_t sz = sel_size(idx[r0 + i], flags[r0 + i], mask, want);
            bytes += sz;
            recs += sz ? 1 : 0;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        bytes += __shfl_xor(bytes, d);
        recs += __shfl_xor(recs, d);
    }
    if ((threadIdx.x & 63u) == 0) { sb[threadIdx.x >> 6] = bytes; sr[threadIdx.x >> 6] = recs; }
    __syncthreads();
    if (threadIdx.x == 0) {
        block_bytes[blockIdx.x] = sb[0] + sb[1] + sb[2] + sb[3];
        block_recs[blockIdx.x] = sr[0] + sr[1] + sr[2] + sr[3];
    }
}

// one block: exclusive scan of block_bytes in place; totals to out[0] (bytes), out[1] (records)
__global__ __launch_bounds__(1024) void k_gather_scan(unsigned long long *__restrict__ block_bytes,
                                                      const unsigned long long *__restrict__ block_recs,
                                                      uint64_t n_blocks, unsigned long long *__restrict__ out) {
    __shared__ unsigned long long wsum[16];
    __shared__ unsigned long long carry, rcarry;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) { carry = 0; rcarry = 0; }
    __syncthreads();
    unsigned long long racc = 0;
    for (uint64_t base = 0; base < n_blocks; base += 1024) {
        const uint64_t i = base + tid;
        const unsigned long long s = i < n_blocks ? block_bytes[i] : 0ull;
        racc += i < n_blocks ? block_recs[i] : 0ull;
        unsigned long long inc = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long o = __shfl_up(inc, d);
            if (lane >= (uint32_t)d) inc += o;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        unsigned long long wbase = carry;
        for (uint32_t w = 0; w < wave; ++w) wbase += wsum[w];
        if (i < n_blocks) block_bytes[i] = wbase + inc - s;
        __syncthreads();
        if (tid == 1023) carry = wbase + inc;
        __syncthreads();
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) racc += __shfl_xor(racc, d);
    if (lane == 0) atomicAdd(&rcarry, racc);
    __syncthreads();
    if (tid == 0) { out[0] = carry; out[1] = rcarry; }
}

// pass 3: a block recomputes its records' exclusive byte prefix, then each wavefront copies the
// selected records of its quarter, one record at a time, 4 bytes per lane and step (source and
// destination are at arbitrary byte offsets: unaligned dword loads and stores, byte tail).
__global__ __launch_bounds__(256) void k_gather_copy(const uint8_t *__restrict__ buf, uint64_t len, uint64_t base_offset,
                                                     const fqh_idx_record *__restrict__ idx, uint64_t n,
                                                     const uint8_t *__restrict__ flags, uint32_t mask, uint32_t want,
                                                     const unsigned long long *__restrict__ block_off,
                                                     uint8_t *__restrict__ out, uint64_t out_cap) {
    __shared__ unsigned long long off[GB_RECS];
    __shared__ uint32_t size[GB_RECS];
    __shared__ unsigned long long wsum[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t r0 = (uint64_t)blockIdx.x * GB_RECS + tid * 4;
    uint32_t sz[4];
    unsigned long long s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        sz[i] = r0 + i < n ? sel_size(idx[r0 + i], flags[r0 + i], mask, want) : 0u;
        s += sz[i];
    }
    unsigned long long inc = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long o = __shfl_up(inc, d);
        if (lane >= (uint32_t)d) inc += o;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned long long ex = block_off[blockIdx.x] + inc - s;
    for (uint32_t w = 0; w < wave; ++w) ex += wsum[w];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        off[tid * 4 + i] = ex;
        size[tid * 4 + i] = sz[i];
        ex += sz[i];
    }
    __syncthreads();
    for (uint32_t k = wave * (GB_RECS / 4); k < (wave + 1) * (GB_RECS / 4); ++k) {
        const uint32_t nbytes = size[k];
        if (!nbytes) continue;  // wave-uniform
        const uint64_t rec = (uint64_t)blockIdx.x * GB_RECS + k;
        const unsigned long long dst = off[k];
        if (dst + nbytes > out_cap) continue;  // the host reports FQH_E_CAPACITY from the totals
        const uint8_t *src = buf + (idx[rec].start - base_offset);
        uint8_t *d = out + dst;
        const uint32_t nd = nbytes >> 2;
        for (uint32_t i = lane; i < nd; i += 64) {
            uint32_t w;
            __builtin_memcpy(&w, src + 4 * i, 4);
            __builtin_memcpy(d + 4 * i, &w, 4);
        }
        if (lane < (nbytes & 3u)) d[4 * nd + lane] = src[4 * nd + lane];
    }
}

void launch_record_flags(hipStream_t s, const uint8_t *buf, uint64_t len, uint64_t base_offset,
                         const fqh_idx_record *idx, uint64_t n, uint8_t *flags) {
    if (!n) return;
    const uint64_t blocks = (n + 31) / 32;
    hipLaunchKernelGGL(k_record_flags, dim3((uint32_t)blocks), dim3(256), 0, s, buf, len, base_offset, idx, n, flags);
}
uint64_t gather_blocks(uint64_t n) { return (n + GB_RECS - 1) / GB_RECS; }
void launch_gather(hipStream_t s, const uint8_t *buf, uint64_t len, uint64_t base_offset, const fqh_idx_record *idx,
                   uint64_t n, const uint8_t *flags, uint32_t mask, uint32_t want, unsigned long long *block_bytes,
                   unsigned long long *block_recs, unsigned long long *totals, uint8_t *out, uint64_t out_cap) {
    const uint64_t nb = gather_blocks(n);
    if (nb) hipLaunchKernelGGL(k_gather_sizes, dim3((uint32_t)nb), dim3(256), 0, s, idx, n, flags, mask, want, block_bytes, block_recs);
    hipLaunchKernelGGL(k_gather_scan, dim3(1), dim3(1024), 0, s, block_bytes, block_recs, nb, totals);
    if (nb && out)
        hipLaunchKernelGGL(k_gather_copy, dim3((uint32_t)nb), dim3(256), 0, s, buf, len, base_offset, idx, n, flags, mask,
                           want, block_bytes, out, out_cap);
}

}  // namespace fqh
