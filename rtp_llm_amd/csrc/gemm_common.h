// Shared pieces of the weight-only GEMM kernels (gemm.hip: staged-x kernels; gemm_smallm.hip: persistent
// x-resident kernel for M <= 8).
#pragma once
#include "common.h"
#include "internal.h"

// One (tile, k-step) unit of the W4 / 64-row path as a fixed instruction stream: the 4 MFMAs of the unit (A = the
// operand dequantised by the previous unit, in the fixed tuple AIN) interleaved with the 13 VALU that dequantise the
// next dword into N0..N3 (the other fixed tuple).  One wave per SIMD issues in order, so the order IS the schedule:
// every MFMA is followed by >= 3 independent VALU (its 16 cycles in the matrix pipe are covered), dependent VALU are
// >= 4 slots apart, and the two leading VALU give the 2 wait states a VALU-written MFMA operand needs (the previous
// unit ends with writes to AIN).  hipcc's own schedule of the same work ran at ~10 cycles per instruction
// (dependent v_pk chains back to back with s_nop between them, accumulators renamed through v_accvgpr moves).
#define WIDE_UNIT_W4(AIN, N0, N1, N2, N3)                                   \
    "v_lshrrev_b32 %[t], 8, %[w]\n\t"                                       \
    "v_and_or_b32 " N0 ", %[w], %[m0], %[e0]\n\t"                           \
    "v_mfma_f32_16x16x32_f16 %[c0], " AIN ", %[b0], %[c0]\n\t"              \
    "v_and_or_b32 " N1 ", %[w], %[m1], %[e1]\n\t"                           \
    "v_and_or_b32 " N2 ", %[t], %[m0], %[e0]\n\t"                           \
    "v_and_or_b32 " N3 ", %[t], %[m1], %[e1]\n\t"                           \
    "v_mfma_f32_16x16x32_f16 %[c1], " AIN ", %[b1], %[c1]\n\t"              \
    "v_pk_add_f16 " N0 ", " N0 ", %[zn]\n\t"                                \
    "v_pk_add_f16 " N1 ", " N1 ", %[znb]\n\t"                               \
    "v_pk_add_f16 " N2 ", " N2 ", %[zn]\n\t"                                \
    "v_mfma_f32_16x16x32_f16 %[c2], " AIN ", %[b2], %[c2]\n\t"              \
    "v_pk_add_f16 " N3 ", " N3 ", %[znb]\n\t"                               \
    "v_pk_mul_f16 " N0 ", " N0 ", %[sc]\n\t"                                \
    "v_pk_mul_f16 " N1 ", " N1 ", %[sc]\n\t"                                \
    "v_mfma_f32_16x16x32_f16 %[c3], " AIN ", %[b3], %[c3]\n\t"              \
    "v_pk_mul_f16 " N2 ", " N2 ", %[sc]\n\t"                                \
    "v_pk_mul_f16 " N3 ", " N3 ", %[sc]"

// Experiment (round 6, VERDICT r05 item 6 iii; tuning build only): the same unit with the wave's priority raised over its MFMA group, so that the SIMD's arbiter
// prefers the wave that is about to feed the matrix pipe over its partner's dequant VALU.
#define WIDE_UNIT_W4_PRIO(AIN, N0, N1, N2, N3)                              \
    "v_lshrrev_b32 %[t], 8, %[w]\n\t"                                       \
    "v_and_or_b32 " N0 ", %[w], %[m0], %[e0]\n\t"                           \
    "s_setprio 2\n\t"                                                       \
    "v_mfma_f32_16x16x32_f16 %[c0], " AIN ", %[b0], %[c0]\n\t"              \
    "v_and_or_b32 " N1 ", %[w], %[m1], %[e1]\n\t"                           \
    "v_and_or_b32 " N2 ", %[t], %[m0], %[e0]\n\t"                           \
    "v_and_or_b32 " N3 ", %[t], %[m1], %[e1]\n\t"                           \
    "v_mfma_f32_16x16x32_f16 %[c1], " AIN ", %[b1], %[c1]\n\t"              \
    "v_pk_add_f16 " N0 ", " N0 ", %[zn]\n\t"                                \
    "v_pk_add_f16 " N1 ", " N1 ", %[znb]\n\t"                               \
    "v_pk_add_f16 " N2 ", " N2 ", %[zn]\n\t"                                \
    "v_mfma_f32_16x16x32_f16 %[c2], " AIN ", %[b2], %[c2]\n\t"              \
    "v_pk_add_f16 " N3 ", " N3 ", %[znb]\n\t"                               \
    "v_pk_mul_f16 " N0 ", " N0 ", %[sc]\n\t"                                \
    "v_pk_mul_f16 " N1 ", " N1 ", %[sc]\n\t"                                \
    "v_mfma_f32_16x16x32_f16 %[c3], " AIN ", %[b3], %[c3]\n\t"              \
    "s_setprio 0\n\t"                                                       \
    "v_pk_mul_f16 " N2 ", " N2 ", %[sc]\n\t"                                \
    "v_pk_mul_f16 " N3 ", " N3 ", %[sc]"

// TIMING ONLY (round 6, tuning build): the unit the route "zero point out of the operand" would run -- 4 v_and_or + shift + 4 v_pk_fma_f16 (1024 + u, s, -1024 s)
// = 9 VALU for the 4 MFMAs.  The operands here are s (u - z) with the zero point applied through the fma's addend (zn / znb times nothing exact): the RESULTS ARE WRONG, the
// instruction stream is what is measured.
#define WIDE_UNIT_W4_FMA9(AIN, N0, N1, N2, N3)                              \
    "v_lshrrev_b32 %[t], 8, %[w]\n\t"                                       \
    "v_and_or_b32 " N0 ", %[w], %[m0], %[e0]\n\t"                           \
    "v_mfma_f32_16x16x32_f16 %[c0], " AIN ", %[b0], %[c0]\n\t"              \
    "v_and_or_b32 " N1 ", %[w], %[m1], %[e1]\n\t"                           \
    "v_and_or_b32 " N2 ", %[t], %[m0], %[e0]\n\t"                           \
    "v_mfma_f32_16x16x32_f16 %[c1], " AIN ", %[b1], %[c1]\n\t"              \
    "v_and_or_b32 " N3 ", %[t], %[m1], %[e1]\n\t"                           \
    "v_pk_fma_f16 " N0 ", " N0 ", %[sc], %[zn]\n\t"                         \
    "v_mfma_f32_16x16x32_f16 %[c2], " AIN ", %[b2], %[c2]\n\t"              \
    "v_pk_fma_f16 " N1 ", " N1 ", %[sc], %[znb]\n\t"                        \
    "v_pk_fma_f16 " N2 ", " N2 ", %[sc], %[zn]\n\t"                         \
    "v_mfma_f32_16x16x32_f16 %[c3], " AIN ", %[b3], %[c3]\n\t"              \
    "v_pk_fma_f16 " N3 ", " N3 ", %[sc], %[znb]"

// Same for 32 rows (2 MFMAs per unit): VALU-issue bound, the MFMAs sit where >= 5 independent VALU follow.
#define WIDE_UNIT_W4_MB2(AIN, N0, N1, N2, N3)                               \
    "v_lshrrev_b32 %[t], 8, %[w]\n\t"                                       \
    "v_and_or_b32 " N0 ", %[w], %[m0], %[e0]\n\t"                           \
    "v_mfma_f32_16x16x32_f16 %[c0], " AIN ", %[b0], %[c0]\n\t"              \
    "v_and_or_b32 " N1 ", %[w], %[m1], %[e1]\n\t"                           \
    "v_and_or_b32 " N2 ", %[t], %[m0], %[e0]\n\t"                           \
    "v_and_or_b32 " N3 ", %[t], %[m1], %[e1]\n\t"                           \
    "v_pk_add_f16 " N0 ", " N0 ", %[zn]\n\t"                                \
    "v_pk_add_f16 " N1 ", " N1 ", %[znb]\n\t"                               \
    "v_mfma_f32_16x16x32_f16 %[c1], " AIN ", %[b1], %[c1]\n\t"              \
    "v_pk_add_f16 " N2 ", " N2 ", %[zn]\n\t"                                \
    "v_pk_add_f16 " N3 ", " N3 ", %[znb]\n\t"                               \
    "v_pk_mul_f16 " N0 ", " N0 ", %[sc]\n\t"                                \
    "v_pk_mul_f16 " N1 ", " N1 ", %[sc]\n\t"                                \
    "v_pk_mul_f16 " N2 ", " N2 ", %[sc]\n\t"                                \
    "v_pk_mul_f16 " N3 ", " N3 ", %[sc]"

// 48 rows (3 MFMAs per unit): as the 64-row unit without its last MFMA.
#define WIDE_UNIT_W4_MB3(AIN, N0, N1, N2, N3)                               \
    "v_lshrrev_b32 %[t], 8, %[w]\n\t"                                       \
    "v_and_or_b32 " N0 ", %[w], %[m0], %[e0]\n\t"                           \
    "v_mfma_f32_16x16x32_f16 %[c0], " AIN ", %[b0], %[c0]\n\t"              \
    "v_and_or_b32 " N1 ", %[w], %[m1], %[e1]\n\t"                           \
    "v_and_or_b32 " N2 ", %[t], %[m0], %[e0]\n\t"                           \
    "v_and_or_b32 " N3 ", %[t], %[m1], %[e1]\n\t"                           \
    "v_pk_add_f16 " N0 ", " N0 ", %[zn]\n\t"                                \
    "v_mfma_f32_16x16x32_f16 %[c1], " AIN ", %[b1], %[c1]\n\t"              \
    "v_pk_add_f16 " N1 ", " N1 ", %[znb]\n\t"                               \
    "v_pk_add_f16 " N2 ", " N2 ", %[zn]\n\t"                                \
    "v_pk_add_f16 " N3 ", " N3 ", %[znb]\n\t"                               \
    "v_pk_mul_f16 " N0 ", " N0 ", %[sc]\n\t"                                \
    "v_mfma_f32_16x16x32_f16 %[c2], " AIN ", %[b2], %[c2]\n\t"              \
    "v_pk_mul_f16 " N1 ", " N1 ", %[sc]\n\t"                                \
    "v_pk_mul_f16 " N2 ", " N2 ", %[sc]\n\t"                                \
    "v_pk_mul_f16 " N3 ", " N3 ", %[sc]"

// 16 rows (1 MFMA per unit): VALU-issue bound; the MFMA sits behind the two leading VALU (the wait states of its VALU-written operand).
#define WIDE_UNIT_W4_MB1(AIN, N0, N1, N2, N3)                               \
    "v_lshrrev_b32 %[t], 8, %[w]\n\t"                                       \
    "v_and_or_b32 " N0 ", %[w], %[m0], %[e0]\n\t"                           \
    "v_mfma_f32_16x16x32_f16 %[c0], " AIN ", %[b0], %[c0]\n\t"              \
    "v_and_or_b32 " N1 ", %[w], %[m1], %[e1]\n\t"                           \
    "v_and_or_b32 " N2 ", %[t], %[m0], %[e0]\n\t"                           \
    "v_and_or_b32 " N3 ", %[t], %[m1], %[e1]\n\t"                           \
    "v_pk_add_f16 " N0 ", " N0 ", %[zn]\n\t"                                \
    "v_pk_add_f16 " N1 ", " N1 ", %[znb]\n\t"                               \
    "v_pk_add_f16 " N2 ", " N2 ", %[zn]\n\t"                                \
    "v_pk_add_f16 " N3 ", " N3 ", %[znb]\n\t"                               \
    "v_pk_mul_f16 " N0 ", " N0 ", %[sc]\n\t"                                \
    "v_pk_mul_f16 " N1 ", " N1 ", %[sc]\n\t"                                \
    "v_pk_mul_f16 " N2 ", " N2 ", %[sc]\n\t"                                \
    "v_pk_mul_f16 " N3 ", " N3 ", %[sc]"

namespace {

struct GemmParams {
    const f16*      x;
    const void*     qw;
    const uint32_t* meta;
    const f16*      bias;
    void*           y;        // direct mode output
    float*          partials; // partial mode output
    int M, K;                 // logical K (row stride of x)
    int N, N_pad, NT, KC;     // NT = N_pad/16, KC = K_pad/128
    int nsplit, cps;          // chunks per split
    int mode;                 // 0 partial slabs, 1 fp16, 2 fp16 silu-mul, 3 fp32
    int ldy;
    int bf16;                 // activations (x, bias, y; W16 weights) are bf16: staged kernel only
    int x_img;                // x is an activation image (common.h act_img_index) of ceil(M / 16) row blocks: wide kernel only
    int y_img;                // fp16 / SiLU-mul outputs are written as an activation image of ceil(M / 16) row blocks (gemm_store)
    uint32_t qw_bytes, meta_bytes, x_bytes;
#ifdef MI355_TUNING
    unsigned long long* stamps;   // tools/wq_stamps.py: wall_clock64 of wave 0 at entry / prologue done / loop done / stores issued / exit, per block
#endif
};

enum { MODE_PARTIAL = 0, MODE_F16 = 1, MODE_SILU = 2, MODE_F32 = 3 };

// voff: per-lane byte offset (VGPR, loop invariant); soff: wave-uniform byte offset (SGPR).  The range check of a
// raw buffer covers voff + soff on gfx950 (tools/probe/soffset_oob.hip), so chunk stepping costs no VALU.
template <int AUX>
__device__ __forceinline__ u32x4 bload128(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff = 0) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX);
}

// ---- in-register widening of the packed codes to MFMA A-operands (no subtract, no scale: both move to
// the accumulator side, see "zero / scale on the C side" below).
//   W4: dword of step s -> 8 fp16:  (e0,e1) = 1024+u  (nibbles at mantissa bits 0-3, exponent of 1024.0)
//                                   (e2,e3) =   64+u  (nibbles at mantissa bits 4-7, exponent of 64.0, ulp 1/16)
//                                   (e4,e5), (e6,e7) the same after one shift by 8.   5 VALU per 8 weights.
//   W8: 8 offset-binary bytes -> 8 fp16 1024+u via v_perm.                              4 VALU per 8 weights.
// v_and_or_b32 is VOP3 (no literal operands on gfx9, one SGPR at most): the four constants live in VGPRs.
struct W4Consts { uint32_t m0, m1, e0, e1; };
__device__ __forceinline__ W4Consts w4_consts() {
    W4Consts c = {0x000F000Fu, 0x00F000F0u, 0x64006400u, 0x54005400u};
    asm volatile("" : "+v"(c.m0), "+v"(c.m1), "+v"(c.e0), "+v"(c.e1)); // keep them in VGPRs
    return c;
}
// One hand-ordered (tile, k-step) unit for MB row blocks (WIDE_UNIT_W4*): the operand of THIS unit sits in the fixed tuple of its
// parity (even units v[100:103] = aE, odd ones v[104:107] = aO) and the dword `wn` is dequantised into the other tuple for the next
// unit.  c0..c3 / b0..b3: accumulators (AGPRs) and B fragments of the unit's row blocks; those >= MB are not touched.
template <int MB, bool EVEN, class BT, int PRIO = 0>   // PRIO: 0 the product stream, 1 s_setprio experiment, 2 the 9-VALU timing-only stream
__device__ __forceinline__ void wide_unit_w4(u32x4& aE, u32x4& aO, uint32_t wn, const W4Consts& k, f16x2 zn, f16x2 znb, f16x2 sc,
                                             f32x4& c0, f32x4& c1, f32x4& c2, f32x4& c3, const BT& b0, const BT& b1, const BT& b2, const BT& b3) {
    uint32_t tmp;
#define WU_IN_  [w] "v"(wn), [m0] "v"(k.m0), [m1] "v"(k.m1), [e0] "v"(k.e0), [e1] "v"(k.e1), [zn] "v"(zn), [znb] "v"(znb), [sc] "v"(sc)
#define WU_EVEN_(MACRO, ACCS, ...) asm volatile(MACRO("v[100:103]", "v104", "v105", "v106", "v107") : [t] "=&v"(tmp), "=&{v[104:107]}"(aO), ACCS : "{v[100:103]}"(aE), WU_IN_, __VA_ARGS__)
#define WU_ODD_(MACRO, ACCS, ...)  asm volatile(MACRO("v[104:107]", "v100", "v101", "v102", "v103") : [t] "=&v"(tmp), "=&{v[100:103]}"(aE), ACCS : "{v[104:107]}"(aO), WU_IN_, __VA_ARGS__)
#define WU_ACC1_ [c0] "+a"(c0)
#define WU_ACC2_ [c0] "+a"(c0), [c1] "+a"(c1)
#define WU_ACC3_ [c0] "+a"(c0), [c1] "+a"(c1), [c2] "+a"(c2)
#define WU_ACC4_ [c0] "+a"(c0), [c1] "+a"(c1), [c2] "+a"(c2), [c3] "+a"(c3)
    if constexpr (MB == 4 && PRIO == 2) {
        if constexpr (EVEN) WU_EVEN_(WIDE_UNIT_W4_FMA9, WU_ACC4_, [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2), [b3] "v"(b3));
        else                WU_ODD_(WIDE_UNIT_W4_FMA9, WU_ACC4_, [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2), [b3] "v"(b3));
    } else if constexpr (MB == 4 && PRIO == 1) {
        if constexpr (EVEN) WU_EVEN_(WIDE_UNIT_W4_PRIO, WU_ACC4_, [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2), [b3] "v"(b3));
        else                WU_ODD_(WIDE_UNIT_W4_PRIO, WU_ACC4_, [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2), [b3] "v"(b3));
    } else if constexpr (MB == 4) {
        if constexpr (EVEN) WU_EVEN_(WIDE_UNIT_W4, WU_ACC4_, [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2), [b3] "v"(b3));
        else                WU_ODD_(WIDE_UNIT_W4, WU_ACC4_, [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2), [b3] "v"(b3));
    } else if constexpr (MB == 3) {
        if constexpr (EVEN) WU_EVEN_(WIDE_UNIT_W4_MB3, WU_ACC3_, [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2));
        else                WU_ODD_(WIDE_UNIT_W4_MB3, WU_ACC3_, [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2));
    } else if constexpr (MB == 2) {
        if constexpr (EVEN) WU_EVEN_(WIDE_UNIT_W4_MB2, WU_ACC2_, [b0] "v"(b0), [b1] "v"(b1));
        else                WU_ODD_(WIDE_UNIT_W4_MB2, WU_ACC2_, [b0] "v"(b0), [b1] "v"(b1));
    } else {
        static_assert(MB == 1, "row blocks: 1..4");
        if constexpr (EVEN) WU_EVEN_(WIDE_UNIT_W4_MB1, WU_ACC1_, [b0] "v"(b0));
        else                WU_ODD_(WIDE_UNIT_W4_MB1, WU_ACC1_, [b0] "v"(b0));
    }
#undef WU_IN_
#undef WU_EVEN_
#undef WU_ODD_
#undef WU_ACC1_
#undef WU_ACC2_
#undef WU_ACC3_
#undef WU_ACC4_
}
__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t m, uint32_t o) {
    return (a & m) | o; // selected as v_and_or_b32 once the constants are opaque VGPRs (no inline asm: its result
                        // feeding an MFMA would need hand-placed wait states)
}
__device__ __forceinline__ f16x8 widen_w4(uint32_t w, const W4Consts& c) {
    const uint32_t w8 = w >> 8;
    u32x4 r;
    r[0] = and_or(w, c.m0, c.e0);
    r[1] = and_or(w, c.m1, c.e1);
    r[2] = and_or(w8, c.m0, c.e0);
    r[3] = and_or(w8, c.m1, c.e1);
    return __builtin_bit_cast(f16x8, r);
}
// bf16: 7 mantissa bits hold a nibble only at the bottom: all four code pairs become 128 + u (0x4300 | u), one bias for the
// whole group, at the price of a shift per pair: 7 VALU per 8 weights.
__device__ __forceinline__ u32x4 widen_w4_bf16(uint32_t w) {
    const uint32_t M = 0x000F000Fu, E = 0x43004300u;
    u32x4 r;
    r[0] = (w & M) | E;
    r[1] = ((w >> 4) & M) | E;
    r[2] = ((w >> 8) & M) | E;
    r[3] = ((w >> 12) & M) | E;
    return r;
}
__device__ __forceinline__ f16x8 widen_w8(uint32_t lo, uint32_t hi) {
    const uint32_t C = 0x64646464u;
    u32x4 r;
    r[0] = __builtin_amdgcn_perm(C, lo, 0x04010400u);
    r[1] = __builtin_amdgcn_perm(C, lo, 0x04030402u);
    r[2] = __builtin_amdgcn_perm(C, hi, 0x04010400u);
    r[3] = __builtin_amdgcn_perm(C, hi, 0x04030402u);
    return __builtin_bit_cast(f16x8, r);
}

// Operand-side dequant (M > 32): scale * (u - z) in fp16 — exact subtract of the biased code, one rounding.
// Even code pairs come out as 1024+u, odd pairs as 64+u (see widen_w4), so the subtract uses two exact
// constants -(1024+z) and -(64+z).  1 shift + 4 v_and_or + 4 v_pk_add + 4 v_pk_mul = 13 VALU per 8 weights.
__device__ __forceinline__ f16x8 dequant_w4_vc(uint32_t w, f16x2 zneg2, f16x2 zneg2b, f16x2 s2, const W4Consts& c) {
    const uint32_t w8 = w >> 8;
    const f16x2 h0 = (as_h2(and_or(w, c.m0, c.e0)) + zneg2) * s2;
    const f16x2 h1 = (as_h2(and_or(w, c.m1, c.e1)) + zneg2b) * s2;
    const f16x2 h2 = (as_h2(and_or(w8, c.m0, c.e0)) + zneg2) * s2;
    const f16x2 h3 = (as_h2(and_or(w8, c.m1, c.e1)) + zneg2b) * s2;
    f16x8 out;
    out[0] = h0[0]; out[1] = h0[1]; out[2] = h1[0]; out[3] = h1[1];
    out[4] = h2[0]; out[5] = h2[1]; out[6] = h3[0]; out[7] = h3[1];
    return out;
}

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}


// Split-K slabs are stored write-through (sc1, 16 bytes per lane through a buffer descriptor over the slab region): see the
// staged kernel's epilogue in gemm.hip for the measurement.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t slab_rsrc(const GemmParams& p, int nsplit) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p.partials, 0, p.mode == MODE_PARTIAL ? (int)((size_t)nsplit * p.M * p.N_pad * 4) : 0, 0x00020000u);
}
__device__ __forceinline__ void st_slab(__amdgpu_buffer_rsrc_t r, uint32_t off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, 16 /*sc1*/);
}

// Epilogue for one lane's 4 consecutive output columns n0..n0+3 of batch row m (v already scaled).
__device__ __forceinline__ void gemm_store(const GemmParams& p, f32x4 v, int m, int n0, int split) {
    if (p.mode == MODE_PARTIAL) {
        st_slab(slab_rsrc(p, p.nsplit), (uint32_t)((((size_t)split * p.M + m) * p.N_pad + n0) * 4), v);
        return;
    }
    if (n0 >= p.N) return;
    if (p.bias) {
        const f16x4 bv = *reinterpret_cast<const f16x4*>(p.bias + n0);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += (float)bv[r];
    }
    if (p.bf16 && p.y_img && p.mode != MODE_F32) {   // bf16 tensors around an fp16 image (wide kernel's image entry; no bias): the values the
        if (p.mode == MODE_F16) {                        // bf16 tensor would hold x 2^-8, stored as fp16 (img_val, common.h)
            f16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (f16)img_val<true>(act_round<true>(v[r]));
            *reinterpret_cast<f16x4*>((f16*)p.y + act_img_index(m, n0, (p.M + 15) >> 4)) = o;
        } else {
            f16x2 o;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float g = act_round<true>(v[2 * t]), u = act_round<true>(v[2 * t + 1]);
                o[t] = (f16)img_val<true>(act_round<true>((g / (1.f + __expf(-g))) * u));   // the SiLU product is where bf16 checkpoints exceed 65504
            }
            *reinterpret_cast<f16x2*>((f16*)p.y + act_img_index(m, n0 >> 1, (p.M + 15) >> 4)) = o;
        }
        return;
    }
    if (p.mode == MODE_F32) {
        *reinterpret_cast<f32x4*>((float*)p.y + (size_t)m * p.ldy + n0) = v;
    } else if (p.mode == MODE_F16) {
        f16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (f16)v[r];
        *reinterpret_cast<f16x4*>((f16*)p.y + (p.y_img ? act_img_index(m, n0, (p.M + 15) >> 4) : (size_t)m * p.ldy + n0)) = o;
    } else { // MODE_SILU: (gate, up) interleaved; round GEMM output to fp16 first
        f16x2 o;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float g = (float)(f16)v[2 * t], u = (float)(f16)v[2 * t + 1];
            o[t] = (f16)((g / (1.f + __expf(-g))) * u);
        }
        // image: the 16 rows of a wave-store are 16 consecutive 16-byte pieces (row-major: 16 separate rows)
        *reinterpret_cast<f16x2*>((f16*)p.y + (p.y_img ? act_img_index(m, n0 >> 1, (p.M + 15) >> 4) : (size_t)m * p.ldy + (n0 >> 1))) = o;
    }
}

} // namespace
