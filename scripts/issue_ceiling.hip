// issue_ceiling.hip — what one SIMD of gfx950 sustains in instructions per cycle at the A* kernel's residency
// (6 single-wave workgroups per SIMD = 24 per CU), for VALU-only, SALU-only and mixed streams of INDEPENDENT
// one-cycle-class integer instructions (v_add_u32 / s_add_u32). VERDICT r1 #7: the "80 % VALU busy" reading of the A*
// profile assumed one quad-cycle per VALU instruction; MI355X_MICROARCH.md says a wave64 VALU op issues over 2 cycles on
// the SIMD-32. This measures it.   build: hipcc --offload-arch=gfx950 -O3 -o issue_ceiling scripts/issue_ceiling.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE> __global__ void __launch_bounds__(64) k(uint32_t* out, int iters) {
    uint32_t v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
    uint32_t s0 = blockIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {   // 128 independent VALU
            REP16(asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n v_add_u32 %4, %4, 1\n v_add_u32 %5, %5, 1\n v_add_u32 %6, %6, 1\n v_add_u32 %7, %7, 1"
                               : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));)
        } else if (MODE == 1) {   // 128 independent SALU
            REP16(asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1"
                               : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");)
        } else {   // 64 VALU + 64 SALU interleaved (the A* kernel's mix is 195 : 153)
            REP16(asm volatile("v_add_u32 %0, %0, 1\n s_add_u32 %4, %4, 1\n v_add_u32 %1, %1, 1\n s_add_u32 %5, %5, 1\n v_add_u32 %2, %2, 1\n s_add_u32 %6, %6, 1\n v_add_u32 %3, %3, 1\n s_add_u32 %7, %7, 1"
                               : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");)
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + s0 + s1 + s2 + s3;
}

template <int MODE> double run(uint32_t* d, int grid, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const double ghz = p.clockRate * 1e-6;
    uint32_t* d;
    hipMalloc(&d, (size_t)cus * 32 * 64 * 4);
    const int iters = 20000;
    printf("# %s, %d CUs, clockRate %.2f GHz (rates are per SIMD per cycle at that clock; 128 instructions per loop iteration + ~3 of loop overhead)\n", p.name, cus, ghz);
    for (int wps : {1, 2, 4, 6, 8}) {   // waves per SIMD
        const int grid = cus * 4 * wps;
        const double simds = cus * 4.0;
        const double n = (double)grid * iters * 128.0;
        const double v = run<0>(d, grid, iters), s = run<1>(d, grid, iters), m = run<2>(d, grid, iters);
        printf("waves/SIMD %d: VALU-only %.3f instr/cycle/SIMD | SALU-only %.3f | mixed 1:1 %.3f (= %.3f VALU + %.3f SALU)\n", wps,
               n / (v * 1e-3) / (ghz * 1e9) / simds, n / (s * 1e-3) / (ghz * 1e9) / simds, n / (m * 1e-3) / (ghz * 1e9) / simds,
               0.5 * n / (m * 1e-3) / (ghz * 1e9) / simds, 0.5 * n / (m * 1e-3) / (ghz * 1e9) / simds);
    }
    hipFree(d);
    return 0;
}
