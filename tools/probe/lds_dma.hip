// Probe: LDS-DMA on gfx950 -- `buffer_load_dwordx4 ... offen lds` with M0 as the LDS destination: layout (lane-linear?),
// reach beyond 64 KB, out-of-range lanes (zeros or untouched?), vmcnt tracking.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void k(const uint32_t* src, int nbytes, uint32_t* out, int lds_off, int oob_from) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    u32x4* dst = reinterpret_cast<u32x4*>(smem + lds_off);
    dst[lane] = (u32x4){0xDEAD0000u + lane, 1, 2, 3};            // sentinel
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    uint32_t voff = lane >= oob_from ? 0x80000000u : (uint32_t)((63 - lane) * 16);   // reversed gather; some lanes out of range
    const uint32_t m0v = (uint32_t)(uintptr_t)dst;                  // LDS byte address of the destination (wave-uniform)
    uint32_t soff = 0;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds\n\ts_waitcnt vmcnt(0)"
                 :: "v"(voff), "s"(r), "s"(__builtin_amdgcn_readfirstlane(m0v)), "s"(soff) : "memory", "m0");
    __syncthreads();
    const u32x4 v = dst[lane];
    out[lane * 4 + 0] = v[0]; out[lane * 4 + 1] = v[1]; out[lane * 4 + 2] = v[2]; out[lane * 4 + 3] = v[3];
}
int main() {
    uint32_t h[256], ho[256], *d, *o;
    for (int i = 0; i < 256; ++i) h[i] = 0xAB000000u + i;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    for (int off : {0, 4096, 70000 & ~15, 130000 & ~15}) {
        k<<<1, 64, 140 * 1024>>>(d, 1024, o, off, 60);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
        printf("lds_off=%6d (%s): lane0 %08x %08x | lane1 %08x | lane59 %08x | lane60 %08x %08x | lane63 %08x\n", off, hipGetErrorString(e),
               ho[0], ho[1], ho[4], ho[59 * 4], ho[60 * 4], ho[60 * 4 + 1], ho[63 * 4]);
    }
    return 0;
}
