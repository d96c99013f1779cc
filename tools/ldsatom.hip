// tools/ldsatom.hip — cost model of ds_add_u32 (LDS atomics without return) on gfx950, the
// primitive behind the histogram kernels (k_stats_lines, k_stats_oct).  One 1024-thread block per CU; every wave issues ITER x 16
// atomics whose address pattern is selected by `pat`.  Prints LDS-array cycles per wave-instruction
// per CU (wall time x clock / instructions issued on that CU).
//   pat 0: lane l -> word l                    (conflict-free, 64 distinct banks/addresses)
//   pat 1: all lanes one address               (64-way same address)
//   pat 2: one row of 64 words, uniform bin in [0,39)   (quality line, current layout)
//   pat 3: 8 copies x 4 of 8 bins, copy = lane&7        (sequence line, current layout)
//   pat 4: row = lane-dependent (64 rows), uniform bin in [0,39)   (lanes on different columns)
//   pat 5: as 2 but 4 distinct bins only       (binned qualities of real instruments)
//   pat 6: as 4 but 4 distinct bins and a per-row rotation of the bin
//   pat 7: as 2 with 2 row copies (copy = lane&1, +64 words)
//   pat 8: as 2 with the bin spread over a 128-word row: word = bin*2 + (lane&1)
//   pat 9: wave-private copy of pattern 0 with `nolds` VALU only (loop overhead reference)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr int ROWS = 128;
template <int PAT>
__global__ __launch_bounds__(1024) void k(unsigned *out, int iters) {
    extern __shared__ unsigned lds[];
    for (unsigned i = threadIdx.x; i < ROWS * 64 * 2; i += 1024) lds[i] = 0;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63u;
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    unsigned r39[16], r4[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {  // per-lane random bins, advanced by +1 (mod) every iteration
        x = x * 1664525u + 1013904223u;
        r39[j] = ((x >> 16) * 39u) >> 16;
        r4[j] = (x >> 20) & 3u;
    }
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned rb = (unsigned)(it * 16) & (ROWS - 1);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            r39[j] = r39[j] == 38u ? 0u : r39[j] + 1u;
            r4[j] = (r4[j] + 1u) & 3u;
            const unsigned row = rb + j;
            unsigned w;
            if (PAT == 0) w = row * 64 + lane;
            else if (PAT == 1) w = row * 64;
            else if (PAT == 2) w = row * 64 + r39[j];
            else if (PAT == 3) w = row * 64 + (lane & 7u) * 8 + (r4[j] * 2 + 1);
            else if (PAT == 4) w = ((row + lane) & (ROWS - 1)) * 64 + r39[j];
            else if (PAT == 5) w = row * 64 + r4[j] * 9;
            else if (PAT == 6) { const unsigned rr = (row + lane) & (ROWS - 1); w = rr * 64 + ((r4[j] * 9 + rr * 5) & 63u); }
            else if (PAT == 7) w = row * 128 + (lane & 1u) * 64 + r39[j];
            else if (PAT == 8) w = row * 128 + r39[j] * 2 + (lane & 1u);
            else if (PAT == 10) w = row * 64 + (lane >> 1);          // 2 lanes per address, 32 banks
            else if (PAT == 11) w = row * 64 + (lane >> 2);          // 4 lanes per address
            else if (PAT == 12) w = row * 64 + (lane & 31u);         // lanes l, l+32 same address
            else if (PAT == 13) w = row * 64 + (lane & 15u) * 2;     // 4 lanes / address, 16 banks
            else { w = 0; acc += r39[j] + r4[j]; }
            if (PAT != 9) atomicAdd(&lds[w], 1u);
        }
    }
    __syncthreads();
    unsigned s = acc;
    for (unsigned i = threadIdx.x; i < ROWS * 64 * 2; i += 1024) s += lds[i];
    if (s == 0x12345678u) out[0] = s;
}
template <int PAT>
void run(unsigned *d, int cus, double mhz) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<PAT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        ROWS * 64 * 2 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int iters = 4000;
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k<PAT>, dim3(cus), dim3(1024), ROWS * 64 * 2 * 4, 0, d, iters);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    const double instr = 16.0 * iters * 16;  // wave-instructions per CU
    printf("pat %2d: %.3f ms  %.2f cycles per wave-instruction per CU (at %.0f MHz)\n", PAT, best,
           best * 1e-3 * mhz * 1e6 / instr, mhz);
}
int main(int argc, char **argv) {
    unsigned *d;
    (void)hipMalloc(&d, 16);
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const double mhz = argc > 1 ? atof(argv[1]) : 2400.0;
    run<9>(d, cus, mhz); run<0>(d, cus, mhz); run<1>(d, cus, mhz); run<12>(d, cus, mhz); run<10>(d, cus, mhz);
    run<11>(d, cus, mhz); run<13>(d, cus, mhz); run<2>(d, cus, mhz); run<3>(d, cus, mhz); run<4>(d, cus, mhz);
    run<5>(d, cus, mhz); run<6>(d, cus, mhz); run<7>(d, cus, mhz); run<8>(d, cus, mhz);
    return 0;
}
