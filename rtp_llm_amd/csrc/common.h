// Shared device/host helpers for libmi355_decode.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <type_traits>
#include "../../include/mi355_decode.h"

typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16   bf16;
typedef __bf16   bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16   bf16x8 __attribute__((ext_vector_type(8)));
typedef float    f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define MI355_WAVE 64

// experiment switches exist only in the tuning build (see gemm.hip); TUNE(i) folds to 0 in the product library
#ifdef MI355_TUNING
extern int g_tune[16];
#define TUNE(i) g_tune[i]
#else
#define TUNE(i) 0
#endif

// host-side error plumbing -------------------------------------------------
void mi355_set_error(const char* fmt, ...);
int  mi355_raise_dynamic_lds(const void* func, const char* name); // once per (kernel, device); error.cpp
#define raise_dynamic_lds mi355_raise_dynamic_lds

#define MI355_CHECK_ARG(cond, ...)            \
    do {                                      \
        if (!(cond)) {                        \
            mi355_set_error(__VA_ARGS__);     \
            return MI355_ERR_ARG;             \
        }                                     \
    } while (0)

#define MI355_CHECK_LAUNCH(name)                                                  \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess) {                                                  \
            mi355_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return MI355_ERR_HIP;                                                 \
        }                                                                         \
    } while (0)

// device helpers -------------------------------------------------------------
__device__ __forceinline__ uint32_t as_u32(f16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ f16x2 as_h2(uint32_t v) { return __builtin_bit_cast(f16x2, v); }

__device__ __forceinline__ f32x4 mfma16x16x32(f16x8 a, f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// ---- activation dtype: kernels that exist in an fp16 and a bf16 form are templated on BF and move 16-bit data as raw dwords
// (two elements each); only the MFMA and the conversions below know the type.  gfx950 has v_mfma_f32_16x16x32_bf16,
// v_dot2_f32_bf16 and v_cvt_pk_bf16_f32 but no packed bf16 add / mul: bf16 arithmetic goes through fp32.
template <bool BF>
__device__ __forceinline__ f32x4 mfma_act(u32x4 a, u32x4 b, f32x4 c) {
    if constexpr (BF) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else              return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <bool BF> __device__ __forceinline__ float act_lo(uint32_t v) {   // element 0 of a packed pair
    if constexpr (BF) return __builtin_bit_cast(float, v << 16);
    else              return (float)__builtin_bit_cast(f16x2, v)[0];
}
template <bool BF> __device__ __forceinline__ float act_hi(uint32_t v) {   // element 1
    if constexpr (BF) return __builtin_bit_cast(float, v & 0xFFFF0000u);
    else              return (float)__builtin_bit_cast(f16x2, v)[1];
}
template <bool BF> __device__ __forceinline__ uint32_t act_pack(float a, float b) {   // round to nearest even, {a, b} -> one dword
    if constexpr (BF) { bf16x2 o; o[0] = (bf16)a; o[1] = (bf16)b; return __builtin_bit_cast(uint32_t, o); }
    else              { f16x2 o;  o[0] = (f16)a;  o[1] = (f16)b;  return __builtin_bit_cast(uint32_t, o); }
}
template <bool BF> __device__ __forceinline__ float act_round(float v) {   // the value a 16-bit tensor of this dtype would hold
    if constexpr (BF) return (float)(bf16)v;
    else              return (float)(f16)v;
}
template <bool BF> __device__ __forceinline__ float act_dot_ones(uint32_t v, float acc) {   // acc + v[0] + v[1]
    if constexpr (BF) { const bf16x2 ones = {(bf16)1.f, (bf16)1.f}; return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, v), ones, acc, false); }
    else              { const f16x2 ones = {(f16)1.f, (f16)1.f};   return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, v), ones, acc, false); }
}

template <bool BF> __device__ __forceinline__ float act_from_bits(uint16_t b) {   // one 16-bit element
    if constexpr (BF) return __builtin_bit_cast(float, (uint32_t)b << 16);
    else              return (float)__builtin_bit_cast(f16, b);
}
template <bool BF> __device__ __forceinline__ uint16_t act_to_bits(float v) {
    if constexpr (BF) return __builtin_bit_cast(uint16_t, (bf16)v);
    else              return __builtin_bit_cast(uint16_t, (f16)v);
}
template <bool BF> __device__ __forceinline__ void act_unpack8(u32x4 v, float (&o)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[2 * i] = act_lo<BF>(v[i]); o[2 * i + 1] = act_hi<BF>(v[i]); }
}
template <bool BF> __device__ __forceinline__ u32x4 act_pack8(const float (&o)[8]) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = act_pack<BF>(o[2 * i], o[2 * i + 1]);
    return v;
}

// 8 unsigned bytes -> 8 bf16 holding the byte values exactly (8 significant bits suffice, but not under a fixed exponent as in
// fp16): v_cvt_f32_ubyteN per byte, then the high halves of two floats packed by one v_perm -- 12 VALU per 8 bytes.  Used for the
// offset-binary INT8 cache (byte = code + 128) and W8 weights (byte = q + 128) under bf16 activations.
__device__ __forceinline__ u32x4 widen_u8_bf16(uint32_t lo, uint32_t hi) {
    auto pk = [](float a, float b) { return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, b), __builtin_bit_cast(uint32_t, a), 0x07060302u); };
    u32x4 r;
    r[0] = pk((float)(lo & 0xFFu), (float)((lo >> 8) & 0xFFu));
    r[1] = pk((float)((lo >> 16) & 0xFFu), (float)(lo >> 24));
    r[2] = pk((float)(hi & 0xFFu), (float)((hi >> 8) & 0xFFu));
    r[3] = pk((float)((hi >> 16) & 0xFFu), (float)(hi >> 24));
    return r;
}

// 8 unsigned nibbles (native W4 order: nibble e at bit 4*(e/2)+16*(e&1)) -> 8 fp16
// holding scale * (u - z) with zneg2 = {-(1024+z)} x2, s2 = {scale} x2.
// (u | 0x6400) is the fp16 1024+u exactly; the add is exact; one rounding in the mul.
__device__ __forceinline__ f16x8 dequant_w4(uint32_t w, f16x2 zneg2, f16x2 s2) {
    const uint32_t M = 0x000F000Fu, E = 0x64006400u;
    f16x2 h0 = as_h2((w & M) | E);
    f16x2 h1 = as_h2(((w >> 4) & M) | E);
    f16x2 h2 = as_h2(((w >> 8) & M) | E);
    f16x2 h3 = as_h2(((w >> 12) & M) | E);
    h0 = (h0 + zneg2) * s2;
    h1 = (h1 + zneg2) * s2;
    h2 = (h2 + zneg2) * s2;
    h3 = (h3 + zneg2) * s2;
    f16x8 r;
    r[0] = h0[0]; r[1] = h0[1]; r[2] = h1[0]; r[3] = h1[1];
    r[4] = h2[0]; r[5] = h2[1]; r[6] = h3[0]; r[7] = h3[1];
    return r;
}

// 8 offset-binary bytes (lo dword = e0..e3, hi dword = e4..e7) -> 8 fp16 (u - z) [* scale]
template <bool SCALE>
__device__ __forceinline__ f16x8 dequant_w8(uint32_t lo, uint32_t hi, f16x2 zneg2, f16x2 s2) {
    // v_perm_b32: bytes of {S0,S1}: selectors 0-3 pick S1 bytes, 4-7 pick S0 bytes.
    const uint32_t C = 0x64646464u;
    f16x2 h0 = as_h2(__builtin_amdgcn_perm(C, lo, 0x04010400u)); // [b0,0x64,b1,0x64]
    f16x2 h1 = as_h2(__builtin_amdgcn_perm(C, lo, 0x04030402u)); // [b2,0x64,b3,0x64]
    f16x2 h2 = as_h2(__builtin_amdgcn_perm(C, hi, 0x04010400u));
    f16x2 h3 = as_h2(__builtin_amdgcn_perm(C, hi, 0x04030402u));
    h0 = h0 + zneg2; h1 = h1 + zneg2; h2 = h2 + zneg2; h3 = h3 + zneg2;
    if (SCALE) { h0 = h0 * s2; h1 = h1 * s2; h2 = h2 * s2; h3 = h3 * s2; }
    f16x8 r;
    r[0] = h0[0]; r[1] = h0[1]; r[2] = h1[0]; r[3] = h1[1];
    r[4] = h2[0]; r[5] = h2[1]; r[6] = h3[0]; r[7] = h3[1];
    return r;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// Combine a value with the lanes 16 / 32 away (lane ^ 16, lane ^ 32) without the LDS crossbar: v_permlane16_swap /
// v_permlane32_swap trade 16-lane rows between two registers (semantics measured on gfx950, tools/probe/permlane_swap.hip:
// 16: a -> [A0 B0 A2 B2], b -> [A1 B1 A3 B3]; 32: a -> [A0 A1 B0 B1], b -> [A2 A3 B2 B3]); with a = b = v the two results
// hold v[lane] and v[lane ^ 16] (resp. ^ 32) in some order, which is all a commutative combine needs.  Pure VALU: no
// lgkmcnt traffic (ds_bpermute, which __shfl_xor compiles to, shares that counter with scalar loads).
// Inline asm, not __builtin_amdgcn_permlane{16,32}_swap: with hipcc 7.2 the builtin's two results came out as ONE register
// whenever they were combined (r[0] + r[1] compiled to v + v; seen in the ISA, caught by the TP engine test).  The leading
// s_nop 1 gives the 2 wait states a VALU-written operand needs before a permlane swap reads it (the compiler's copy of v
// into the second register sits right in front of the statement and hipcc pads nothing inside or before an asm).
__device__ __forceinline__ void swap16(float v, float& r0, float& r1) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    r0 = a; r1 = b;
}
__device__ __forceinline__ void swap32(float v, float& r0, float& r1) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    r0 = a; r1 = b;
}
__device__ __forceinline__ float xor16_max(float v) { float a, b; swap16(v, a, b); return fmaxf(a, b); }
__device__ __forceinline__ float xor32_max(float v) { float a, b; swap32(v, a, b); return fmaxf(a, b); }
__device__ __forceinline__ float xor16_sum(float v) { float a, b; swap16(v, a, b); return a + b; }
__device__ __forceinline__ float xor32_sum(float v) { float a, b; swap32(v, a, b); return a + b; }

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// the same with the dtype as a run-time flag: epilogues that run once per output element and are not worth a second kernel instance
__device__ __forceinline__ float rt_from_bits(uint16_t b, bool bf) { return bf ? act_from_bits<true>(b) : act_from_bits<false>(b); }
__device__ __forceinline__ uint16_t rt_to_bits(float v, bool bf) { return bf ? act_to_bits<true>(v) : act_to_bits<false>(v); }
__device__ __forceinline__ float rt_round(float v, bool bf) { return bf ? act_round<true>(v) : act_round<false>(v); }

// ---- activation image (mi355_act_image_*): the [M][K] 16-bit activations of a 5-64-row step in the order the full-K launches of
// gemm_fullk64.hip read them -- one dense 1 KB run per MFMA B fragment (k-step of 32, row block of 16): lane (jj, q) of the
// fragment owns the 16 bytes x[16 rb + jj][32 ks + 8 q .. + 7].  A fragment gathered from the row-major tensor instead is 16 runs
// of 64 B at the row stride: with every CU of the chip asking for the same runs, the L2 serves them at ~7 TB/s against 35 for dense
// runs (profiles/r04_fullk64_access_patterns.txt: 19.8 vs 11.2 us for the QKV launch at 64 rows).  The producers (RMSNorm,
// attention) write this order directly: their 16-byte stores keep their size, only the address changes.
// An image ALWAYS holds fp16: the GEMMs that read it run fp16 MFMAs on operand-side dequantised weights.  A bf16 step converts
// on the way in (img_pack8 below).  bf16 reaches 3.4e38, fp16 65504 -- and the SiLU * up product of a bf16 checkpoint is where
// "massive activations" live -- so the image of a BF16 tensor holds fp16(x 2^-8) (kImgBfScale: exact, a power of two), saturated at
// +-65504, and every GEMM that reads such an image multiplies its fp32 accumulators by 2^8 (kImgBfUnscale; gemm_fullk64 / gemm_wide /
// gemm_splitk64, exact).  Range: |x| < 1.6e7.  Precision: elements below 2^-6 land in fp16's subnormals and keep an ABSOLUTE
// spacing of 2^-16 -- finer than bf16's own spacing down to |x| = 2^-8, and below that 2^-17 absolute next to a row's typical
// element.  (The deferred-norm image gamma 2^-e h' carries its own exponent, e + 8 for bf16: engine.cpp.)  fp16 steps: unscaled.
constexpr float kImgBfScale = 0.00390625f, kImgBfUnscale = 256.f;
// the fp16 an image stores for element v (already rounded to the BF-typed tensor it stands for)
template <bool BF> __device__ __forceinline__ float img_val(float v) {
    if constexpr (!BF) return v;
    const float s = v * kImgBfScale;
    return s != s ? s : __builtin_amdgcn_fmed3f(s, -65504.f, 65504.f);   // saturate, but a NaN stays a NaN (v_med3 would return a bound: ADVICE r05)
}
__device__ __forceinline__ float rt_img_val(float v, bool bf) { return bf ? img_val<true>(v) : v; }
// Element index of x[row][col]; mblk = row blocks of the image = ceil(M / 16).
// 8 values of a BF-typed tensor (already rounded to it) -> the 16 bytes an image stores
template <bool BF> __device__ __forceinline__ u32x4 img_pack8(const float (&o)[8]) {
    if constexpr (!BF) return act_pack8<false>(o);
    float r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = img_val<true>(act_round<true>(o[i]));
    return act_pack8<false>(r);
}
__host__ __device__ __forceinline__ size_t act_img_index(int row, int col, int mblk) {
    return ((((size_t)(col >> 5) * mblk + (row >> 4)) * 64 + ((col & 31) >> 3) * 16 + (row & 15)) << 3) + (col & 7);
}

// compile-time loop: the body sees its index as a constant expression (std::integral_constant)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
