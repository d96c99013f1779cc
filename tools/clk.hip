// tools/clk.hip — effective shader clock under load: every wave runs a dependent integer chain of
// known length; cycles = s_memtime delta, wall = hipEvents.  Prints MHz while all CUs are busy.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(unsigned long long *out, int iters) {
    unsigned x = threadIdx.x;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) { x = x * 1664525u + 1013904223u; x ^= x >> 7; }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (x == 12345) out[1] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
int main() {
    unsigned long long *d, h[2];
    hipMalloc(&d, 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k, dim3(256 * 8), dim3(256), 0, 0, d, 2000000);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("run %d: %.2f ms, %llu cycles -> %.0f MHz\n", r, ms, h[0], h[0] / ms / 1e3);
    }
    return 0;
}
