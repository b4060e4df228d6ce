// Probe: semantics of v_permlane16_swap_b32 / v_permlane32_swap_b32 on gfx950 (which 16-lane rows of the two registers trade places).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;                 // a: rows A0..A3 = lanes 0-15, 16-31, ...; b: B0..B3
    unsigned a16 = a, b16 = b, a32 = a, b32 = b, ac = a, bc = b;
    asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a16), "+v"(b16));
    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a32), "+v"(b32));
    asm volatile("v_permlane16_swap_b32 %0, %1\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(ac), "+v"(bc));   // the combination the GEMM wants
    out[threadIdx.x * 6 + 0] = a16; out[threadIdx.x * 6 + 1] = b16; out[threadIdx.x * 6 + 2] = a32; out[threadIdx.x * 6 + 3] = b32;
    out[threadIdx.x * 6 + 4] = ac; out[threadIdx.x * 6 + 5] = bc;
}
int main() {
    unsigned* d; unsigned h[64 * 6];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[6] = {"permlane16_swap a", "permlane16_swap b", "permlane32_swap a", "permlane32_swap b", "16 then 32   a", "16 then 32   b"};
    for (int v = 0; v < 6; ++v) {
        printf("%-18s rows:", names[v]);
        for (int r = 0; r < 4; ++r) { unsigned x = h[(r * 16) * 6 + v]; printf("  %c%u", x >= 100 ? 'B' : 'A', (x % 100) / 16); }
        printf("   (lane 0 of each row: %u %u %u %u)\n", h[0 * 6 + v], h[16 * 6 + v], h[32 * 6 + v], h[48 * 6 + v]);
    }
    return 0;
}
