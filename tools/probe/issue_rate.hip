// Probe: shader clock under load, VALU issue rate per SIMD, and MFMA/VALU co-issue on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE> // 0: VALU only, 1: MFMA only, 2: MFMA + 4 VALU interleaved, 3: MFMA + 2 VALU
__global__ __launch_bounds__(1024) void k(unsigned* out, unsigned long long* clk, int iters) {
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, m = 0x000F000F + blockIdx.x, o = 0x64006400;
    asm volatile("" : "+v"(m), "+v"(o));
    f16x8 fa, fb; f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 0.001f); fb[i] = (_Float16)0.5f; }
    unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE != 0) {
                if ((u & 3) == 0) c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, c0, 0, 0, 0);
                if ((u & 3) == 1) c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, c1, 0, 0, 0);
                if ((u & 3) == 2) c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, c2, 0, 0, 0);
                if ((u & 3) == 3) c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, c3, 0, 0, 0);
            }
            if (MODE == 0 || MODE == 2) { a0 = (a0 & m) | o; a1 = (a1 & m) | o; a2 = (a2 & m) | o; a3 = (a3 & m) | o; a0 += a3; a1 += a0; a2 += a1; a3 += a2; }
            if (MODE == 3) { a0 = (a0 & m) | o; a1 = (a1 & m) | o; a0 += a1; a1 += a0; }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    out[blockIdx.x * 1024 + threadIdx.x] = a0 + a1 + a2 + a3 + (unsigned)(c0[0] + c1[1] + c2[2] + c3[3]);
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

int main() {
    unsigned* out; unsigned long long *clk, h[2];
    hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&clk, 16);
    const int iters = 20000;
    for (int threads = 256; threads <= 1024; threads *= 2) {       // ONE block per CU, 1/2/4 waves per SIMD inside it
        const int blocks = 256;
        for (int mode = 0; mode < 4; ++mode) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (mode == 0) k<0><<<blocks, threads>>>(out, clk, iters);
            if (mode == 1) k<1><<<blocks, threads>>>(out, clk, iters);
            if (mode == 2) k<2><<<blocks, threads>>>(out, clk, iters);
            if (mode == 3) k<3><<<blocks, threads>>>(out, clk, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
            printf("block=%d threads (%d waves/SIMD) mode=%d: kernel %.3f ms, block0 %.3f ms, cycles per body per wave %.1f, clock %.2f GHz\n",
                   threads, threads / 256, mode, ms, h[1] / 1e5, h[0] / (iters * 8.0), h[0] / (h[1] * 10.0) / 1e3 * 1e3 / 1e3);
        }
    }
    return 0;
}
