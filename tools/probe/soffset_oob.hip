// Probe: does the buffer range check include the SGPR offset (soffset) on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned* buf, unsigned* out, int soff) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, 1024, 0x00020000);
    u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 16, soff, 0);          // soffset path
    u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 16 + soff, 0, 0);      // voffset path
    out[threadIdx.x * 2] = a[0]; out[threadIdx.x * 2 + 1] = b[0];
}
int main() {
    unsigned *d, *o, h[2048], ho[128];
    for (int i = 0; i < 2048; ++i) h[i] = 0xAB000000u + i;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (int soff : {0, 512, 1024, 2048}) {
        k<<<1, 64>>>(d, o, soff); hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
        printf("soff=%d: lane0 soffset-path=%08x voffset-path=%08x | lane40 %08x %08x\n", soff, ho[0], ho[1], ho[80], ho[81]);
    }
    return 0;
}
