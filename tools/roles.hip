// tools/roles.hip — would a producer / consumer split of k_scan_stats' block pay?  A model of one CU's work per 4 KiB group: the SCAN part
// (4 ds_write_b128, 4 ds_read_b128, ~200 dependent VALU, two more dependent LDS round trips) and the COUNT part (4 rounds of: a
// descriptor read, 8 ds_read2_b32 that depend on it, ~60 VALU, 16 atomics).  mode 0: sixteen wavefronts do scan + count of their own
// group one after the other (today's kernel).  mode 1: eight wavefronts only scan (two groups per turn), eight only count (two per
// turn): the same instructions per CU, no hand-off cost at all — an upper bound of what specialisation can give.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) unsigned lds_u32;
__device__ __forceinline__ unsigned scan_part(unsigned char *area, unsigned lane, uint4 &a, uint4 &b, uint4 &c, uint4 &d) {
    *reinterpret_cast<uint4 *>(area + 16 * lane) = a;
    *reinterpret_cast<uint4 *>(area + 1024 + 16 * lane) = b;
    *reinterpret_cast<uint4 *>(area + 2048 + 16 * lane) = c;
    *reinterpret_cast<uint4 *>(area + 3072 + 16 * lane) = d;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint4 r0 = *reinterpret_cast<const uint4 *>(area + 64 * lane), r1 = *reinterpret_cast<const uint4 *>(area + 64 * lane + 16);
    const uint4 r2 = *reinterpret_cast<const uint4 *>(area + 64 * lane + 32), r3 = *reinterpret_cast<const uint4 *>(area + 64 * lane + 48);
    unsigned x = 0;
    const unsigned w[16] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x, r3.y, r3.z, r3.w};
#pragma unroll
    for (int i = 0; i < 16; ++i) {  // ~10 dependent-ish VALU per dword
        unsigned t = ((w[i] & 0x7F7F7F7Fu) ^ 0x0A0A0A0Au) + 0x7F7F7F7Fu;
        t = ~(t | w[i]) & 0x80808080u;
        x = x * 33u + __builtin_amdgcn_udot4(t, 0x08040201u, x, false);
        x ^= x >> 7; x += t >> 3; x ^= w[i] << 1; x += x >> 11; x *= 5u;
    }
    unsigned short *lst = reinterpret_cast<unsigned short *>(area + 4096);
    lst[lane] = (unsigned short)x;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    unsigned e = lst[(lane + 1) & 63u];
    unsigned by = area[(e * 7u) & 4095u];
    x += by;
    a.x += x; b.y ^= x; c.z += x; d.w ^= x;
    return x;
}
__device__ __forceinline__ unsigned count_part(unsigned char *area, unsigned *hist, unsigned lane, unsigned seed) {
    unsigned x = seed;
    const unsigned *desc = reinterpret_cast<const unsigned *>(area + 4096);
#pragma unroll 1
    for (int r = 0; r < 4; ++r) {
        const unsigned dsc = desc[(lane >> 3) + 8 * r] + x;            // descriptor round trip
        const unsigned base = (dsc * 13u) & 0xFFCu;
        unsigned w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {  // line reads that depend on it
            const unsigned p = (base + 32u * u + 4u * (lane & 7u)) & 4088u;
            const unsigned lo = *reinterpret_cast<const unsigned *>(area + p), hi = *reinterpret_cast<const unsigned *>(area + p + 4);
            w[u] = __builtin_amdgcn_alignbyte(hi, lo, dsc & 3u);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // 16 atomics per round, bank = lane
            const unsigned t = w[u] - 0x21212121u + w[u + 4];
            x |= t & 0xC0C0C0C0u;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned addr = (((t >> (8 * k)) & 63u) << 8) + ((lane & 63u) << 2);
                (void)__hip_atomic_fetch_add((lds_u32 *)(unsigned long long)(unsigned)(unsigned long long)(hist + (addr >> 2)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    return x;
}
template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned *hist = reinterpret_cast<unsigned *>(lds);            // 16 KiB
    for (unsigned i = threadIdx.x; i < 4096; i += 1024) hist[i] = 0;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    unsigned char *area = lds + 16384 + wv * 6144;
    uint4 a = make_uint4(threadIdx.x, 2, 3, 4), b = a, c = a, d = a;
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            acc += scan_part(area, lane, a, b, c, d);
            acc += count_part(area, hist, lane, acc);
        } else if (wv < 8) {
            acc += scan_part(area, lane, a, b, c, d);
            acc += scan_part(area, lane, a, b, c, d);
        } else {
            acc += count_part(area, hist, lane, acc);
            acc += count_part(area, hist, lane, acc);
        }
    }
    __syncthreads();
    if (acc == 0x12345 || hist[threadIdx.x] == 0xFFFFFFFFu) out[0] = acc;
}
int main(int argc, char **argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 2000;
    unsigned *out; hipMalloc(&out, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = 16384 + 16 * 6144;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 3; ++rep)
        for (int mode = 0; mode < 2; ++mode) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(1024), lds, 0, out, iters);
            else hipLaunchKernelGGL(k<1>, dim3(256), dim3(1024), lds, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("mode %d (%s): %.3f ms for %d turns of 16 groups per CU = %.1f ns per group and CU\n", mode, mode ? "8 scanners + 8 counters" : "16 unified", ms, iters, ms * 1e6 / (iters * 16.0));
        }
    return 0;
}
