// How many single-wave workgroups of a given dynamic-LDS size are resident per CU? Time a fixed loop at grid =
// n_cu * k for k = 16..32: the time doubles when the grid no longer fits in one round.
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ unsigned smem[];
__global__ void __launch_bounds__(64, 6) probe(unsigned long long* out, int iters) {
    unsigned x = threadIdx.x;
    smem[threadIdx.x] = x;
    for (int i = 0; i < iters; ++i) x = x * 1664525u + smem[(x >> 7) & 63u];
    if (x == 0x12345) out[0] = x;
}
int main() {
    unsigned long long* d;
    hipMalloc(&d, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int lds : {4096, 5632, 6144, 6656, 7168, 8192}) {
        printf("LDS %5d B:", lds);
        for (int k : {16, 19, 20, 21, 22, 23, 24, 25, 28, 32}) {
            hipLaunchKernelGGL(probe, dim3(256 * k), dim3(64), lds, 0, d, 200000);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(probe, dim3(256 * k), dim3(64), lds, 0, d, 200000);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("  k=%d %.1fms", k, ms);
        }
        printf("\n");
    }
    return 0;
}
