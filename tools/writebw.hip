// tools/writebw.hip — HBM store ceiling for the shape k_emit writes: N u64 values, one per lane
// (512 contiguous bytes per wave-store), against 16 bytes per lane, on a persistent grid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int W>
__global__ __launch_bounds__(256) void k(unsigned long long *out, size_t n, int chunk) {
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nt = (size_t)gridDim.x * 256;
    if (W == 8) {
        if (chunk == 0) {
            for (size_t i = tid; i < n; i += nt) out[i] = i;
        } else {  // a wave writes `chunk` values (like one tile's record starts), then jumps
            const size_t lane = threadIdx.x & 63, wave = tid >> 6, nw = nt >> 6;
            const size_t nchunks = n / chunk;
            for (size_t c = wave; c < nchunks; c += nw)
                if (lane < (size_t)chunk) out[c * chunk + lane] = c;
        }
    } else {
        uint4 *o = reinterpret_cast<uint4 *>(out);
        for (size_t i = tid; i < n / 2; i += nt) o[i] = make_uint4((unsigned)i, 0, (unsigned)i, 1);
    }
}
int main(int argc, char **argv) {
    const size_t n = 52060209;
    unsigned long long *d;
    (void)hipMalloc(&d, n * 8 + 64);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (int blocks : {256, 512, 1024, 2048, 4096, 16384}) {
        for (int mode = 0; mode < 4; ++mode) {
            float best = 1e9f;
            for (int r = 0; r < 5; ++r) {
                (void)hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(256), 0, 0, d, n, 0);
                else if (mode == 1) hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(256), 0, 0, d, n, 50);
                else if (mode == 2) hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(256), 0, 0, d, n, 64);
                else hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(256), 0, 0, d, n, 0);
                (void)hipEventRecord(b);
                (void)hipEventSynchronize(b);
                float ms = 0;
                (void)hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            printf("blocks %5d %-22s %.3f ms  %.0f GB/s\n", blocks,
                   mode == 0 ? "u64/lane grid-stride" : mode == 1 ? "u64/lane 50 per wave" : mode == 2 ? "u64/lane 64 per wave" : "16 B/lane grid-stride",
                   best, n * 8 / 1e6 / best);
        }
    }
    return 0;
}
