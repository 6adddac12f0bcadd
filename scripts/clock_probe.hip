// Calibrates the shader clock: clock64() (s_memtime) against wall_clock64() (constant-rate counter) for a lone wave
// and for a fully occupied chip running a VALU+SALU loop.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned long long* out, int iters) {
    unsigned long long c0 = clock64(), w0 = wall_clock64();
    unsigned x = threadIdx.x, s = blockIdx.x;
    for (int i = 0; i < iters; ++i) {
        x = x * 1664525u + 1013904223u;
        s = s * 22695477u + 1u;
        x ^= __builtin_amdgcn_readfirstlane(s);
    }
    unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[blockIdx.x * 3] = c1 - c0; out[blockIdx.x * 3 + 1] = w1 - w0; out[blockIdx.x * 3 + 2] = x; }
}
int main() {
    int rate_khz = 0;
    hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    printf("wall clock rate %d kHz, max engine clock %d kHz\n", rate_khz, clk_khz);
    unsigned long long* d;
    hipMalloc(&d, 8192 * 3 * 8);
    for (int grid : {1, 256, 6144}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(probe, dim3(grid), dim3(64), 0, 0, d, 20000000 / (grid > 256 ? 4 : 1));
            hipDeviceSynchronize();
        }
        unsigned long long h[3];
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        double secs = (double)h[1] / (rate_khz * 1e3);
        printf("grid %5d: clock64 %llu ticks, wall %llu ticks = %.3f ms -> clock64 rate %.1f MHz\n", grid, h[0], h[1], secs * 1e3, h[0] / secs / 1e6);
    }
    return 0;
}
