// tools/ldpat.hip — what a CU's L1 (TA/TCP) makes of the load shapes a line-per-8-lanes kernel can use.
// Every wave walks "lines" of a 16 KiB window (stride 330 bytes, 8 lines per instruction, 8 lanes per
// line) that stays in cache; prints cycles per wave instruction per CU and bytes per cycle per CU.
//   0: dword, 64 lanes contiguous, aligned (reference)
//   1: dword,   8 lanes x 4 B per line, line base 4-byte aligned
//   2: dword,   8 lanes x 4 B per line, line base odd
//   3: dwordx2, 8 lanes x 8 B per line, line base odd
//   4: dwordx4, 8 lanes x 16 B per line, line base odd
//   5: dwordx4, 8 lanes x 16 B per line, line base 16-byte aligned
//   6: dwordx2, 8 lanes x 8 B per line, line base 8-byte aligned
//   7: dwordx4, 64 lanes contiguous, aligned (the scan kernel's shape)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int PAT>
__global__ __launch_bounds__(1024) void k(const unsigned char *buf, unsigned *out, int iters) {
    const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6, g = lane >> 3, m = lane & 7u;
    const unsigned char *base = buf + (size_t)blockIdx.x * 16 * 32768 + (wv & 0u);  // one 13 KiB window per CU: L1 hits
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned rec = ((unsigned)it * 8u + g) % 40u;  // 40 * 330 < 16 KiB
        unsigned off = rec * 330u;
        if (PAT == 1) off &= ~3u;
        if (PAT == 2 || PAT == 3 || PAT == 4) off |= 1u;
        if (PAT == 5) off &= ~15u;
        if (PAT == 6) off &= ~7u;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (PAT == 0) {
                unsigned v; __builtin_memcpy(&v, base + (((unsigned)it * 4u + u) % 60u) * 256u + lane * 4u, 4); acc += v;
            } else if (PAT == 1 || PAT == 2) {
                unsigned v; __builtin_memcpy(&v, base + off + m * 4u + 32u * u, 4); acc += v;
            } else if (PAT == 3 || PAT == 6) {
                uint2 v; __builtin_memcpy(&v, base + off + m * 8u + 64u * u, 8); acc += v.x ^ v.y;
            } else if (PAT == 4 || PAT == 5) {
                uint4 v; __builtin_memcpy(&v, base + off + m * 16u + 128u * (u & 1), 16); acc += v.x ^ v.y ^ v.z ^ v.w;
            } else {
                uint4 v; __builtin_memcpy(&v, base + (((unsigned)it * 4u + u) % 15u) * 1024u + lane * 16u, 16); acc += v.x ^ v.y ^ v.z ^ v.w;
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int PAT>
void run(const unsigned char *buf, unsigned *d, int cus, double mhz, int bytes) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int iters = 2000;
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k<PAT>, dim3(cus), dim3(1024), 0, 0, buf, d, iters);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    const double instr = 16.0 * iters * 4;  // wave instructions per CU
    const double cyc = best * 1e-3 * mhz * 1e6 / instr;
    printf("pat %d: %.3f ms  %.1f cycles per wave instruction per CU, %.1f useful bytes per cycle per CU\n", PAT, best, cyc,
           bytes / cyc);
}
int main(int argc, char **argv) {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const double mhz = argc > 1 ? atof(argv[1]) : 2100.0;
    unsigned char *buf;
    unsigned *d;
    (void)hipMalloc(&buf, (size_t)cus * 16 * 32768 + 4096);
    (void)hipMemset(buf, 1, (size_t)cus * 16 * 32768 + 4096);
    (void)hipMalloc(&d, 16);
    run<0>(buf, d, cus, mhz, 256); run<1>(buf, d, cus, mhz, 256); run<2>(buf, d, cus, mhz, 256);
    run<3>(buf, d, cus, mhz, 512); run<6>(buf, d, cus, mhz, 512); run<4>(buf, d, cus, mhz, 1024);
    run<5>(buf, d, cus, mhz, 1024); run<7>(buf, d, cus, mhz, 1024);
    return 0;
}
