// Which SIMD does wave w of a workgroup land on?  (HW_REG_HW_ID: wave_id [3:0], simd_id [5:4], cu_id [11:8], sh_id [12], se_id [15:13])
// usage: hipcc --offload-arch=gfx950 -O2 -o wave_simd wave_simd.hip && ./wave_simd [threads_per_block]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k(unsigned* out) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = hw;
}
int main(int argc, char** argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 768, nb = 8;
    unsigned* d; hipMalloc(&d, nb * 16 * 4); hipMemset(d, 0xff, nb * 16 * 4);
    hipLaunchKernelGGL(k, dim3(nb), dim3(threads), 0, 0, d);
    unsigned h[8 * 16]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < nb; ++b) {
        printf("block %d (%d waves): simd of wave 0..: ", b, threads / 64);
        for (int w = 0; w < threads / 64; ++w) printf("%u ", (h[b * 16 + w] >> 4) & 3);
        printf("  cu %u\n", (h[b * 16] >> 8) & 15);
    }
    return 0;
}
