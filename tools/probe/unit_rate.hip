// Probe: cost of one gemm_wide (tile, k-step) unit on gfx950 -- 4 MFMA 16x16x32 f16 + 13 dequant VALU in a fixed order --
// against its parts, with 1 and 2 waves per SIMD.  Prints shader cycles per unit per wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define V1 "v_lshrrev_b32 %[t], 8, %[w]\n\t"
#define V2(N0) "v_and_or_b32 " N0 ", %[w], %[m0], %[e0]\n\t"
#define MF(C, AIN, B) "v_mfma_f32_16x16x32_f16 " C ", " AIN ", " B ", " C "\n\t"
#define UNIT(AIN, N0, N1, N2, N3, M0, M1, M2, M3)              \
    V1 V2(N0) M0                                                \
    "v_and_or_b32 " N1 ", %[w], %[m1], %[e1]\n\t"               \
    "v_and_or_b32 " N2 ", %[t], %[m0], %[e0]\n\t"               \
    "v_and_or_b32 " N3 ", %[t], %[m1], %[e1]\n\t" M1            \
    "v_pk_add_f16 " N0 ", " N0 ", %[zn]\n\t"                    \
    "v_pk_add_f16 " N1 ", " N1 ", %[znb]\n\t"                   \
    "v_pk_add_f16 " N2 ", " N2 ", %[zn]\n\t" M2                 \
    "v_pk_add_f16 " N3 ", " N3 ", %[znb]\n\t"                   \
    "v_pk_mul_f16 " N0 ", " N0 ", %[sc]\n\t"                    \
    "v_pk_mul_f16 " N1 ", " N1 ", %[sc]\n\t" M3                 \
    "v_pk_mul_f16 " N2 ", " N2 ", %[sc]\n\t"                    \
    "v_pk_mul_f16 " N3 ", " N3 ", %[sc]"

// MODE 0: full unit; 1: VALU only; 2: MFMA only; 3: 2 x 32x32x16 + 13 VALU; 4: full unit, MFMAs back to back at the end;
// 5: what a 32x32x16 kernel would really issue for TWO units (a tile pair x 32 k): v_permlane16_swap + v_permlane32_swap of the
//    two packed code dwords (tools/probe/permlane_swap.hip), 2 x 13 dequant VALU, 4 x 32x32x16 -- printed per unit (half of it)
template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void k(unsigned* out, unsigned long long* clk, int iters) {
    unsigned w = threadIdx.x * 2654435761u, m0 = 0x000F000F, m1 = 0x00F000F0, e0 = 0x64006400, e1 = 0x54005400;
    unsigned zn = 0xE408E408, znb = 0xD440D440, sc = 0x20002000, t, w2 = w ^ 0x9E3779B9u;
    asm volatile("" : "+v"(m0), "+v"(m1), "+v"(e0), "+v"(e1), "+v"(zn), "+v"(znb), "+v"(sc), "+v"(w), "+v"(w2));
    f16x8 b0, b1, b2, b3;
    for (int i = 0; i < 8; ++i) { b0[i] = (_Float16)(threadIdx.x * 0.001f); b1[i] = (_Float16)0.5f; b2[i] = (_Float16)0.25f; b3[i] = (_Float16)0.125f; }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    f32x16 d0, d1;
    for (int i = 0; i < 16; ++i) { d0[i] = 0; d1[i] = 0; }
    u32x4 aE = {e0, e0, e0, e0}, aO = aE;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#define ARGS_E : [t] "=&v"(t), "=&{v[44:47]}"(aO), [c0] "+a"(c0), [c1] "+a"(c1), [c2] "+a"(c2), [c3] "+a"(c3), [d0] "+a"(d0), [d1] "+a"(d1) \
               : "{v[40:43]}"(aE), [w] "v"(w), [m0] "v"(m0), [m1] "v"(m1), [e0] "v"(e0), [e1] "v"(e1), [zn] "v"(zn), [znb] "v"(znb), [sc] "v"(sc), \
                 [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2), [b3] "v"(b3)
#define ARGS_O : [t] "=&v"(t), "=&{v[40:43]}"(aE), [c0] "+a"(c0), [c1] "+a"(c1), [c2] "+a"(c2), [c3] "+a"(c3), [d0] "+a"(d0), [d1] "+a"(d1) \
               : "{v[44:47]}"(aO), [w] "v"(w), [m0] "v"(m0), [m1] "v"(m1), [e0] "v"(e0), [e1] "v"(e1), [zn] "v"(zn), [znb] "v"(znb), [sc] "v"(sc), \
                 [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2), [b3] "v"(b3)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 0) {
                asm volatile(UNIT("v[40:43]", "v44", "v45", "v46", "v47", MF("%[c0]", "v[40:43]", "%[b0]"), MF("%[c1]", "v[40:43]", "%[b1]"),
                                  MF("%[c2]", "v[40:43]", "%[b2]"), MF("%[c3]", "v[40:43]", "%[b3]")) ARGS_E);
                asm volatile(UNIT("v[44:47]", "v40", "v41", "v42", "v43", MF("%[c0]", "v[44:47]", "%[b0]"), MF("%[c1]", "v[44:47]", "%[b1]"),
                                  MF("%[c2]", "v[44:47]", "%[b2]"), MF("%[c3]", "v[44:47]", "%[b3]")) ARGS_O);
            } else if (MODE == 1) {
                asm volatile(UNIT("v[40:43]", "v44", "v45", "v46", "v47", "", "", "", "") ARGS_E);
                asm volatile(UNIT("v[44:47]", "v40", "v41", "v42", "v43", "", "", "", "") ARGS_O);
            } else if (MODE == 2) {
                asm volatile(MF("%[c0]", "v[40:43]", "%[b0]") MF("%[c1]", "v[40:43]", "%[b1]") MF("%[c2]", "v[40:43]", "%[b2]") MF("%[c3]", "v[40:43]", "%[b3]") "s_nop 0" ARGS_E);
                asm volatile(MF("%[c0]", "v[44:47]", "%[b0]") MF("%[c1]", "v[44:47]", "%[b1]") MF("%[c2]", "v[44:47]", "%[b2]") MF("%[c3]", "v[44:47]", "%[b3]") "s_nop 0" ARGS_O);
            } else if (MODE == 3) {
#define MF32(C, AIN, B) "v_mfma_f32_32x32x16_f16 " C ", " AIN ", " B ", " C "\n\t"
                asm volatile(UNIT("v[40:43]", "v44", "v45", "v46", "v47", MF32("%[d0]", "v[40:43]", "%[b0]"), "", MF32("%[d1]", "v[40:43]", "%[b1]"), "") ARGS_E);
                asm volatile(UNIT("v[44:47]", "v40", "v41", "v42", "v43", MF32("%[d0]", "v[44:47]", "%[b0]"), "", MF32("%[d1]", "v[44:47]", "%[b1]"), "") ARGS_O);
            } else if (MODE == 5) {
#define SWAPS "v_permlane16_swap_b32 %[w], %[w2]\n\ts_nop 1\n\tv_permlane32_swap_b32 %[w], %[w2]\n\t"
#define UNIT2(W, AIN, N0, N1, N2, N3, M0, M1)                       \
    "v_lshrrev_b32 %[t], 8, " W "\n\t"                             \
    "v_and_or_b32 " N0 ", " W ", %[m0], %[e0]\n\t" M0               \
    "v_and_or_b32 " N1 ", " W ", %[m1], %[e1]\n\t"                  \
    "v_and_or_b32 " N2 ", %[t], %[m0], %[e0]\n\t"                   \
    "v_and_or_b32 " N3 ", %[t], %[m1], %[e1]\n\t"                   \
    "v_pk_add_f16 " N0 ", " N0 ", %[zn]\n\t"                        \
    "v_pk_add_f16 " N1 ", " N1 ", %[znb]\n\t" M1                    \
    "v_pk_add_f16 " N2 ", " N2 ", %[zn]\n\t"                        \
    "v_pk_add_f16 " N3 ", " N3 ", %[znb]\n\t"                       \
    "v_pk_mul_f16 " N0 ", " N0 ", %[sc]\n\t"                        \
    "v_pk_mul_f16 " N1 ", " N1 ", %[sc]\n\t"                        \
    "v_pk_mul_f16 " N2 ", " N2 ", %[sc]\n\t"                        \
    "v_pk_mul_f16 " N3 ", " N3 ", %[sc]\n\t"
                // operands of this pair-step in v[40:43] (k 0..15) and v[44:47] (k 16..31); the next pair-step's go to v[48:55]
                asm volatile(SWAPS
                             UNIT2("%[w]", "", "v48", "v49", "v50", "v51", MF32("%[d0]", "v[40:43]", "%[b0]"), MF32("%[d1]", "v[40:43]", "%[b1]"))
                             UNIT2("%[w2]", "", "v52", "v53", "v54", "v55", MF32("%[d0]", "v[44:47]", "%[b2]"), MF32("%[d1]", "v[44:47]", "%[b3]"))
                             "v_mov_b32 v40, v48\n\tv_mov_b32 v44, v52"     // stands for the alternation of two fixed tuples (free in the real stream)
                             : [t] "=&v"(t), "+{v[40:43]}"(aE), "+{v[44:47]}"(aO), [d0] "+a"(d0), [d1] "+a"(d1), [w] "+v"(w), [w2] "+v"(w2)
                             : [m0] "v"(m0), [m1] "v"(m1), [e0] "v"(e0), [e1] "v"(e1), [zn] "v"(zn), [znb] "v"(znb), [sc] "v"(sc),
                               [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2), [b3] "v"(b3)
                             : "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
            } else {
                asm volatile(UNIT("v[40:43]", "v44", "v45", "v46", "v47", "", "", "", "") "\n\t"
                             MF("%[c0]", "v[40:43]", "%[b0]") MF("%[c1]", "v[40:43]", "%[b1]") MF("%[c2]", "v[40:43]", "%[b2]") MF("%[c3]", "v[40:43]", "%[b3]") "s_nop 0" ARGS_E);
                asm volatile(UNIT("v[44:47]", "v40", "v41", "v42", "v43", "", "", "", "") "\n\t"
                             MF("%[c0]", "v[44:47]", "%[b0]") MF("%[c1]", "v[44:47]", "%[b1]") MF("%[c2]", "v[44:47]", "%[b2]") MF("%[c3]", "v[44:47]", "%[b3]") "s_nop 0" ARGS_O);
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * THREADS + threadIdx.x] = aE[0] + aO[1] + t + w2 + (unsigned)(c0[0] + c1[1] + c2[2] + c3[3] + d0[0] + d1[5]);
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int MODE, int THREADS>
void run(unsigned* out, unsigned long long* clk) {
    const int iters = 5000;
    unsigned long long h;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, THREADS><<<256, THREADS>>>(out, clk, iters);
    hipEventRecord(e0);
    k<MODE, THREADS><<<256, THREADS>>>(out, clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    printf("waves/SIMD=%d mode=%d: %.3f ms, %.1f shader cycles per unit per wave (%.1f ns wall per unit)\n", THREADS / 256, MODE, ms,
           h / (iters * 8.0), ms * 1e6 / (iters * 8.0));
}

int main() {
    unsigned* out; unsigned long long* clk;
    hipMalloc(&out, 512 * 256 * 4); hipMalloc(&clk, 16);
    run<0, 256>(out, clk); run<1, 256>(out, clk); run<2, 256>(out, clk); run<3, 256>(out, clk); run<4, 256>(out, clk); run<5, 256>(out, clk);
    run<0, 512>(out, clk); run<1, 512>(out, clk); run<2, 512>(out, clk); run<3, 512>(out, clk); run<4, 512>(out, clk); run<5, 512>(out, clk);
    return 0;
}
