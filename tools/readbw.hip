// tools/readbw.hip — microbenchmark: which streaming-read pattern reaches the best HBM read rate on
// MI355X?  (Feeds DESIGN.md's "measured streaming-read ceiling".)  Build: hipcc --offload-arch=gfx950
// -O3 -o tools/bin/readbw tools/readbw.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <bool NT> __device__ __forceinline__ uint4 ld(const uint4* p) {
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    if (NT) { u4 r = __builtin_nontemporal_load((const u4*)p); return make_uint4(r.x, r.y, r.z, r.w); }
    return *p;
}
__device__ __forceinline__ unsigned long long sum4(uint4 v) { return (unsigned long long)v.x + v.y + v.z + v.w; }

// wave-tile pattern: wave owns TILE bytes contiguous, reads 1 KiB pieces, U loads in flight
template <int U, bool NT, int TILE>
__global__ __launch_bounds__(256) void k_wavetile(const uint8_t* buf, uint64_t len, unsigned long long* out) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t n_tiles = len / TILE;
    const uint64_t nw = (uint64_t)gridDim.x * 4;
    unsigned long long acc = 0;
    for (uint64_t t = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < n_tiles; t += nw) {
        const uint8_t* base = buf + t * TILE + lane * 16;
#pragma unroll 1
        for (int g = 0; g < TILE / 1024 / U; ++g) {
            uint4 v[U];
#pragma unroll
            for (int j = 0; j < U; ++j) v[j] = ld<NT>((const uint4*)(base + (g * U + j) * 1024));
#pragma unroll
            for (int j = 0; j < U; ++j) acc += sum4(v[j]);
        }
    }
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    if (lane == 0 && acc) atomicAdd(out, acc);
}
// block-contiguous pattern: a block reads BLK*16 bytes per load instruction, U in flight
template <int U, bool NT, int BLK>
__global__ __launch_bounds__(BLK) void k_blockrow(const uint8_t* buf, uint64_t len, unsigned long long* out) {
    const uint64_t chunk = (uint64_t)BLK * 16 * U;
    const uint64_t n = len / chunk;
    unsigned long long acc = 0;
    for (uint64_t c = blockIdx.x; c < n; c += gridDim.x) {
        const uint8_t* base = buf + c * chunk + threadIdx.x * 16;
        uint4 v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) v[j] = ld<NT>((const uint4*)(base + (uint64_t)j * BLK * 16));
#pragma unroll
        for (int j = 0; j < U; ++j) acc += sum4(v[j]);
    }
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}


// LDS-DMA pattern (global_load_lds_dwordx4, gfx950): a wave owns TILE contiguous bytes and streams them into its own LDS
// ring, U KiB per round, two rounds in flight; one ds_read_b128 per lane and round keeps the data "used".  Inline asm: the
// compiler brackets its own LDS-DMA builtin with vmcnt(0) in front of every LDS read, which would serialise the rounds.
template <int U, bool NT, int TILE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_dma(const uint8_t* buf, uint64_t len, unsigned long long* out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t base = (uint32_t)(uintptr_t)lds + w * (2 * U * 1024);
    const uint64_t n_tiles = len / TILE;
    const uint64_t nw = (uint64_t)gridDim.x * WAVES;
    uint32_t acc = 0;
    constexpr int ROUNDS = TILE / 1024 / U;
    auto issue = [&](const uint8_t* p, uint32_t ldsb) {
        const uint32_t m0v = __builtin_amdgcn_readfirstlane(ldsb);
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const uint8_t* q = p + j * 1024;
            if (NT) asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off nt" :: "s"(m0v + j * 1024), "v"(q) : "memory", "m0");
            else asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(m0v + j * 1024), "v"(q) : "memory", "m0");
        }
    };
    uint64_t t = (uint64_t)blockIdx.x * WAVES + w;
    if (t >= n_tiles) return;
    issue(buf + t * TILE + lane * 16, base);
    uint32_t par = 0;
    for (; t < n_tiles; t += nw) {
        const uint64_t tn = t + nw < n_tiles ? t + nw : t;
#pragma unroll 1
        for (int g = 0; g < ROUNDS; ++g) {
            const uint8_t* nx = (g + 1 < ROUNDS) ? buf + t * TILE + (g + 1) * U * 1024 + lane * 16 : buf + tn * TILE + lane * 16;
            issue(nx, base + (par ^ 1) * U * 1024);
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(U) : "memory");
            const uint4 v = *reinterpret_cast<const uint4*>(lds + (base - (uint32_t)(uintptr_t)lds) + par * U * 1024 + lane * 16);
            acc += v.x ^ v.y ^ v.z ^ v.w;
            par ^= 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long a = acc;
    for (int d = 32; d >= 1; d >>= 1) a += __shfl_xor(a, d);
    if (lane == 0 && a == 0x123456789ull) atomicAdd(out, a);
}

template <typename F> float timeit(F f, int reps = 7) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); best = std::min(best, ms);
    }
    return best;
}

int main(int argc, char** argv) {
    uint64_t len = argc > 1 ? strtoull(argv[1], 0, 0) : (16ull << 30);
    uint8_t* buf; unsigned long long* out;
    CK(hipMalloc(&buf, len)); CK(hipMalloc(&out, 8));
    CK(hipMemset(buf, 1, len)); CK(hipMemset(out, 0, 8));
    auto rep = [&](const char* name, float ms) { printf("%-44s %8.3f ms %8.1f GB/s %5.1f%%\n", name, ms, len / 1e6 / ms, len / 1e6 / ms / 80.0); fflush(stdout); };
#define WT(U, NT, TILE, GRID) rep("wavetile U=" #U " nt=" #NT " tile=" #TILE " grid=" #GRID, timeit([&] { hipLaunchKernelGGL((k_wavetile<U, NT, TILE>), dim3(GRID), dim3(256), 0, 0, buf, len, out); }))
#define BR(U, NT, BLK, GRID) rep("blockrow U=" #U " nt=" #NT " blk=" #BLK " grid=" #GRID, timeit([&] { hipLaunchKernelGGL((k_blockrow<U, NT, BLK>), dim3(GRID), dim3(BLK), 0, 0, buf, len, out); }))
#define DM(U, NT, TILE, WAVES, BPC) rep("lds-dma U=" #U " nt=" #NT " tile=" #TILE " waves/blk=" #WAVES " blk/cu=" #BPC, timeit([&] { hipLaunchKernelGGL((k_dma<U, NT, TILE, WAVES>), dim3(256 * BPC), dim3(WAVES * 64), WAVES * 2 * U * 1024, 0, buf, len, out); }))
    DM(4, true, 16384, 4, 4);
    DM(4, false, 16384, 4, 4);
    DM(4, true, 16384, 4, 3);
    DM(4, true, 16384, 4, 2);
    DM(8, true, 16384, 4, 2);
    DM(2, true, 16384, 4, 4);
    DM(4, true, 65536, 4, 4);
    DM(4, true, 16384, 16, 1);
    WT(4, false, 16384, 2048);
    WT(4, false, 16384, 4096);
    WT(4, false, 16384, 262144);
    WT(8, false, 16384, 2048);
    WT(16, false, 16384, 2048);
    WT(4, true, 16384, 2048);
    WT(8, true, 16384, 2048);
    WT(8, true, 16384, 1024);
    WT(8, false, 65536, 2048);
    BR(4, false, 256, 2048);
    BR(8, false, 256, 2048);
    BR(8, false, 256, 4096);
    BR(8, true, 256, 2048);
    BR(4, false, 1024, 512);
    BR(8, false, 1024, 512);
    BR(4, true, 1024, 512);
    BR(4, false, 512, 1024);
    BR(16, false, 256, 2048);
    BR(8, false, 256, 1048576);
    BR(4, false, 256, 1048576);
    return 0;
}
