// Paged flash-decoding attention (GQA) with fp16 or INT8 KV-cache, gfx950: q_len = 1 (decode) and q_len > 1 rows per
// sequence with the causal mask applied inside the page walk (speculative target-verify, `is_target_verify`
// bindings/OpDefs.h:283, and chunked prefill over the paged cache, FusedRopeKVCacheOp.cc:216-461 / aiter.py:244-950).
//
// Replaces AiterDecodeAttnOp*.forward / paged_attention_atrex
// (rtp_llm/models_py/modules/factory/attention/rocm_impl/aiter.py:1340-1561,
// rtp_llm/models_py/bindings/rocm/atrexPA.cc:444-496); numerics follow the
// reference's torch oracle run_native/ref_masked_attention
// (modules/base/rocm/test/rocm_fmha_test.py:262-372): fp32 logits
// scale*q.k (* k_scale), fp32 softmax, (* v_scale), P.V.
//
// Work split: grid (partition, kv_head, sequence); a block of 4 waves owns one
// partition of the sequence, each wave streams 32-token groups and keeps its own
// online-softmax state (wavefront split-K); waves merge through LDS, partitions
// through a small reduce kernel.  All G = nh/nkv query heads of a kv head ride
// in the 16-wide N dimension of v_mfma_f32_16x16x32_f16, so K/V are read once
// per kv head:
//   S^T[tok][j] : A = K tile (16 tokens x 32 d),   B = q^T (32 d x 16 heads)
//   O^T[d][j]   : A = V^T tile (16 d x 32 tokens), B = P   (32 tokens x 16 heads)
// The two 16-token K tiles of a group take tokens {8w+r} and {8w+4+r} (w = lane>>4)
// so that the S accumulators of a lane *are* its P.V B-fragment (8 consecutive
// tokens) — no cross-lane traffic between the two MFMA chains.  V is stored
// channel-major per block so the V^T fragment is one 16-byte load.
// INT8: K/V bytes are widened in-register (v_perm + exact fp16 add), the per-token
// scales are applied to S (K) and to P (V) in fp32.
// Multi-row: the 16 MFMA columns of a tile hold (row, head) pairs -- R = 16 / G rows of one sequence x its G query heads --
// and a block carries NT such tiles against the same K/V fragments, so a sequence's KV is streamed once per NT*R query rows
// instead of once per row; column j masks tokens beyond ITS row's position (causality), nothing else changes.
#include <type_traits>

#include "common.h"
#include "internal.h"

namespace {

struct AttnParams {
    const f16*     q;
    const void*    kv_base;
    const float*   scale_base;
    const int32_t* block_table;
    const int32_t* seq_lens;
    f16*           out;
    float*         tmp_out; // [B][nh][P][hd]
    float*         tmp_ml;  // [B][nh][P][2]
    int B, nh, nkv, G, page, max_blocks, P, PS, seq_add, max_seq, num_blocks;
    int q_len, R, ntile;   // rows per sequence, rows per 16-column tile (16 / G), row tiles per sequence
    uint32_t page_magic;   // 2^32 / page + 1: floor(t / page) == (t * magic) >> 32 for t * page < 2^32
    float scale_log2; // softmax scale * log2(e)
    int img_mblk;     // > 0: out is an activation image of that many row blocks (common.h act_img_index) instead of [rows][nh * hd]
#ifdef MI355_TUNING
    unsigned long long* stamps;   // tools/attn_stamps.py: wall_clock64 per wave at entry / first requests out / first group computed / loop done / merged / exit
#endif
};
#ifdef MI355_TUNING
#define AT_STAMP(i) do { if (p.stamps && lane == 0) p.stamps[((((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + wave) * 6 + (i)] = wall_clock64(); } while (0)
#else
#define AT_STAMP(i) do { } while (0)
#endif
// element index of out[row][col], col = h * HD + d
__device__ __forceinline__ size_t out_index(const AttnParams& p, int row, int col, int HD) {
    return p.img_mblk > 0 ? act_img_index(row, col, p.img_mblk) : (size_t)row * p.nh * HD + col;
}

constexpr float NEG_BIG = -1e30f;

// NW waves per block split the partition between them (each wave keeps two 32-token groups of K/V in flight).  NW = 8 for
// the one-block-per-CU grids (B * nkv * P <= 256) was measured and lost: 34.2 vs 29.8 us at b = 64 / ctx 1024 fp16 KV and
// 98 vs 89 us at ctx 4096 INT8 KV -- two waves per SIMD cap the kernel at 256 registers and it spills 12-21 of them.
// 8 cache bytes (code + 128) -> 8 fp16 holding 1152 + k (offset-binary byte under the exponent of 1024: exact).  The bias is NOT
// subtracted per element: it rides through the MFMA and leaves as one term per output, 1152 * sum_d q (scores) or
// 1152 * sum_tok p (values) -- 4 VALU per 8 bytes instead of 10, on a path that is VALU-issue bound.
__device__ __forceinline__ f16x8 widen_kv8(uint32_t lo, uint32_t hi) {
    const uint32_t C = 0x64646464u;   // the cache holds code + 128 (rope_kv.hip): no sign fix-up here
    u32x4 r;
    r[0] = __builtin_amdgcn_perm(C, lo, 0x04010400u);
    r[1] = __builtin_amdgcn_perm(C, lo, 0x04030402u);
    r[2] = __builtin_amdgcn_perm(C, hi, 0x04010400u);
    r[3] = __builtin_amdgcn_perm(C, hi, 0x04030402u);
    return __builtin_bit_cast(f16x8, r);
}

// BF: Q, a 16-bit cache and the output are bf16 (kv_dtype MI355_KV_BF16): bf16 MFMAs for S = K q^T and O = V^T P, P rounded to
// bf16.  The INT8 cache serves both activation dtypes: its bytes are widened to fp16 (widen_kv8, bias 1152) or to bf16
// (widen_u8_bf16, bias 128).
template <int HD, bool INT8, int NT, int NW, int NG, bool BF = false>
__global__ __launch_bounds__(64 * NW) void paged_attn_kernel(const AttnParams p) {
    constexpr float KVB = BF ? 128.f : 1152.f;     // INT8: what a widened cache byte carries on top of its code (see widen_kv8 / _bf16)
    constexpr int NTHR = 64 * NW, GS = 32 * NW;    // threads; tokens one round of the block's waves covers
    constexpr int NSTEP = HD / 32; // QK k-steps
    constexpr int NDB   = HD / 16; // PV d-blocks
    const int part = blockIdx.x, kh = blockIdx.y;
    const int b = blockIdx.z / p.ntile, tile = blockIdx.z - b * p.ntile;
    const int RT = NT * p.R;                       // query rows of this block
    const int row0 = b * p.q_len + tile * RT;      // first row (index into q / positions / out)
    const int nrows = min(RT, p.q_len - tile * RT);
    // context of a row = its position + 1 (seq_add = 1: positions hold tokens already cached) or seq_lens[row] itself;
    // a negative value marks a padding row.  The block walks up to the longest context of its rows.
    const int pstart = part * p.PS;
    // the block ids of every wave's first group are asked for TOGETHER with the context lengths (scalar loads, one round trip):
    // their table slots depend on the partition only.  Slots past a short context hold whatever the table holds -- the ids are
    // clamped where they are used and those tokens carry p = 0.  (They used to wait for seq_len: a second dependent round trip
    // in front of the first K/V request; tools/attn_stamps.py.)
    const int32_t* bt = p.block_table + (size_t)b * p.max_blocks;
    const int wave_pre = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int first_ids[4];
    {
        const int tb0 = pstart + wave_pre * 32, tmax = p.max_blocks * p.page - 1;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int wt = min(tb0 + 8 * a, tmax);
            first_ids[a] = bt[(int)__builtin_amdgcn_readfirstlane((int)(((unsigned long long)(unsigned)wt * p.page_magic) >> 32))];
        }
    }
    int seq_len = 0;
    for (int i = 0; i < nrows; ++i) seq_len = max(seq_len, min(p.seq_lens[row0 + i] + p.seq_add, p.max_seq));
    if (seq_len <= 0) {   // only padding rows here: their outputs are defined (zeros), nothing is read
        if (part == 0)
            for (int idx = threadIdx.x; idx < nrows * p.G * HD; idx += NTHR) {
                const int rl = idx / (p.G * HD), rem = idx - rl * (p.G * HD);
                p.out[out_index(p, row0 + rl, kh * p.G * HD + rem, HD)] = (f16)0.f;
            }
        return;
    }
    if (pstart >= seq_len) return;
    const int pend = min(seq_len, pstart + p.PS);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, w = lane >> 4;
    AT_STAMP(0);

    // column j of tile c = (row c*R + j/G, head kh*G + j%G); q fragments (B operand), zero for unused columns
    u32x4 qf[NT][NSTEP];
    int   limit[NT];       // tokens [0, limit) are visible to this lane's column (causal mask); 0 = nothing
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        const int rl = c * p.R + j / p.G, g = j - (j / p.G) * p.G;
        const bool ok = j < p.R * p.G && rl < nrows;
        const int row = row0 + (ok ? rl : 0);
        limit[c] = ok ? min(max(p.seq_lens[row] + p.seq_add, 0), p.max_seq) : 0;
        const f16* qrow = p.q + ((size_t)row * p.nh + kh * p.G + g) * HD;
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            const int d = INT8 ? (s >> 1) * 64 + w * 16 + (s & 1) * 8 : s * 32 + w * 8;
            u32x4 v = *reinterpret_cast<const u32x4*>(qrow + d);
            if (!ok) v = (u32x4){0u, 0u, 0u, 0u};
            qf[c][s] = v;
        }
    }

    // INT8: 1152 * sum_d q of the lane's column (fp32; the S tile of a lane is one column), see widen_kv8
    float kq[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        kq[c] = 0.f;
        if (INT8) {
            const f16x2 ones = {(f16)1.f, (f16)1.f};
            float qs = 0.f;
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                const u32x4 v = qf[c][s];
#pragma unroll
                for (int e = 0; e < 4; ++e) qs = act_dot_ones<BF>(v[e], qs);
            }
            qs = xor32_sum(xor16_sum(qs));
            kq[c] = KVB * qs;
        }
    }
    const size_t head_elems = (size_t)p.page * HD;
    const uint64_t v_off = (uint64_t)p.nkv * head_elems;   // V heads sit nkv heads behind the K heads of a block
    const char* kvb = (const char*)p.kv_base;
    constexpr int ES = INT8 ? 1 : 2;

    f32x4 o[NT][NDB];
    float m_run[NT], l_run[NT];
    float l16_run[NT];     // INT8: running sum of the fp16-rounded, V-scaled probabilities the PV MFMA actually consumed
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        m_run[c] = NEG_BIG; l_run[c] = 0.f; l16_run[c] = 0.f;
#pragma unroll
        for (int db = 0; db < NDB; ++db) o[c][db] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    // One 32-token group of K/V in registers (+ the per-token INT8 scales of the lane's own 8 tokens).
    struct Group {
        u32x4 kf[2][INT8 ? NSTEP / 2 : NSTEP];
        u32x4 vf16[INT8 ? 1 : NDB];
        u32x2 vf8[INT8 ? NDB : 1];
        f32x4 ksc[2], vsc[2];
    };
    const int last = seq_len - 1;
    // Page addressing is WAVE-UNIFORM work: a 32-token group is four 8-token windows, each inside one page (page % 8 == 0).
    // The block ids of a group are fetched by SCALAR loads (s_load, lgkmcnt) and turned into the four windows' element
    // offsets on the scalar unit; a lane then only selects the window of its K rows (j >> 2) and of its V / scale / P
    // slots (w).  Vector loads here (round 2) were waited for with vmcnt, an IN-ORDER counter: asking for the ids of group
    // n + 2 drained every K/V load of group n + 1 issued just before, so a wave never had a group in flight while it
    // computed (one 16 KB group per ~3.5 us of loaded latency per wave = 4.4 TB/s at b = 64 / ctx 1024).
    struct Wins { int raw[4]; };                       // block ids as loaded (unclamped; clamped where they are used)
    const int lastw = last & ~7;                       // whole windows are clamped in range (their tokens carry p = 0)
    // window token -> (page index, token inside the page): floor(t / page) as one s_mul_hi with magic = 2^32 / page + 1,
    // exact for t * page < 2^32 (checked on the host)
    auto page_of = [&](int wt) { return (int)__builtin_amdgcn_readfirstlane((int)(((unsigned long long)(unsigned)wt * p.page_magic) >> 32)); };
    auto lookup = [&](int tb, Wins& wn) {
#pragma unroll
        for (int a = 0; a < 4; ++a) wn.raw[a] = bt[page_of(min(tb + 8 * a, lastw))];
    };
    // the first group's ids came with the prologue's scalar loads; a window past the context takes the id of the last window, as lookup does
    auto first_wins = [&](Wins& wn) {
        const int tb0 = pstart + wave * 32;
#pragma unroll
        for (int a = 0; a < 4; ++a) wn.raw[a] = first_ids[a];
        if (tb0 + 24 > lastw) lookup(tb0, wn);           // the context ends inside this group (rare: the last group of a sequence)
    };
    // lane -> window selectors, loop invariant
    const bool k1 = (j >> 2) == 1, k2 = (j >> 2) == 2, k3 = (j >> 2) == 3;
    const bool v1 = w == 1, v2 = w == 2, v3 = w == 3;
    auto load_group = [&](Group& g, int tb, const Wins& wn) {
        long kof[4], vof[4]; int sof[4];               // element offsets of the windows (uniform: scalar unit)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int wt = min(tb + 8 * a, lastw);
            const int ti = wt - page_of(wt) * p.page;
            const int blk = min(max(wn.raw[a], 0), p.num_blocks - 1);
            // 32 x 32 -> 64-bit products (s_mul_i32 + s_mul_hi_u32): the head index fits 31 bits (host check on the scale plane), a
            // head's elements 32; with `long` operands the scalar unit ran a 64 x 64 multiply chain per window (~40 instructions)
            const uint32_t hk = (uint32_t)blk * (uint32_t)(2 * p.nkv) + (uint32_t)kh;
            const uint64_t kbase = (uint64_t)hk * (uint64_t)(uint32_t)head_elems;
            kof[a] = (long)(kbase + (uint32_t)(ti * HD));
            vof[a] = (long)(kbase + v_off + (uint32_t)ti);
            sof[a] = (int)(hk * (uint32_t)p.page) + ti;
        }
        long kb = kof[0]; kb = k1 ? kof[1] : kb; kb = k2 ? kof[2] : kb; kb = k3 ? kof[3] : kb;
        long vb = vof[0]; vb = v1 ? vof[1] : vb; vb = v2 ? vof[2] : vb; vb = v3 ? vof[3] : vb;
        // K: tile tau row j -> token (j & 3) + 4 tau of window j >> 2
        const char* kl = kvb + (kb + (long)((j & 3) * HD)) * ES + (INT8 ? w * 16 : w * 16);
#pragma unroll
        for (int tau = 0; tau < 2; ++tau) {
            const char* krow = kl + tau * 4 * HD * ES;
            if (INT8) {
#pragma unroll
                for (int sp = 0; sp < NSTEP / 2; ++sp) g.kf[tau][sp] = *reinterpret_cast<const u32x4*>(krow + sp * 64);
            } else {
#pragma unroll
                for (int s = 0; s < NSTEP; ++s) g.kf[tau][s] = *reinterpret_cast<const u32x4*>(krow + s * 64);
            }
        }
        // V: d-block db row j (channel db*16+j), the 8 tokens of window w
        const char* vl = kvb + (vb + (long)j * p.page) * ES;
        const long vstep = (long)16 * p.page * ES;
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            const char* vrow = vl + db * vstep;
            if (INT8) g.vf8[db] = *reinterpret_cast<const u32x2*>(vrow);
            else      g.vf16[db] = *reinterpret_cast<const u32x4*>(vrow);
        }
        if (INT8) { // scale plane [blk][K|V][nkv][page]: the lane's S rows / P slots are the 8 tokens of window w
            int so = sof[0]; so = v1 ? sof[1] : so; so = v2 ? sof[2] : so; so = v3 ? sof[3] : so;
            const float* ks = p.scale_base + so;
            const float* vs = ks + (size_t)p.nkv * p.page;
            g.ksc[0] = *reinterpret_cast<const f32x4*>(ks); g.ksc[1] = *reinterpret_cast<const f32x4*>(ks + 4);
            g.vsc[0] = *reinterpret_cast<const f32x4*>(vs); g.vsc[1] = *reinterpret_cast<const f32x4*>(vs + 4);
        }
    };
    auto compute_group = [&](const Group& g, int tb, auto masked_c) {
        constexpr bool MASKED = decltype(masked_c)::value;   // false: every token of the group is visible to every column
        const int vwin = tb + w * 8; // first token of this lane's S rows / P slots
        // ---- widen K once, S^T = K q^T for every column tile
        u32x4 ka[2][NSTEP];
#pragma unroll
        for (int tau = 0; tau < 2; ++tau)
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                if (INT8) {
                    const u32x4 kk = g.kf[tau][s >> 1];
                    if constexpr (BF) ka[tau][s] = widen_u8_bf16(kk[(s & 1) * 2], kk[(s & 1) * 2 + 1]);
                    else              ka[tau][s] = __builtin_bit_cast(u32x4, widen_kv8(kk[(s & 1) * 2], kk[(s & 1) * 2 + 1]));
                } else {
                    ka[tau][s] = g.kf[tau][s];
                }
            }
        u32x4 pf[NT];
        float alpha[NT];
        bool rescale = false;
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            f32x4 sacc[2];
#pragma unroll
            for (int tau = 0; tau < 2; ++tau) {
                sacc[tau] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < NSTEP; ++s) sacc[tau] = mfma_act<BF>(ka[tau][s], qf[c][s], sacc[tau]);
            }
            // ---- scale, causal mask, online softmax (log2 domain)
            float sv[8];
            float mx = NEG_BIG;
#pragma unroll
            for (int tau = 0; tau < 2; ++tau)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = (INT8 ? sacc[tau][r] - kq[c] : sacc[tau][r]) * p.scale_log2;
                    if (INT8) v *= g.ksc[tau][r];
                    const int tok = vwin + tau * 4 + r;
                    if (MASKED) v = tok < limit[c] ? v : NEG_BIG;
                    sv[tau * 4 + r] = v;
                    mx = fmaxf(mx, v);
                }
            mx = xor32_max(xor16_max(mx));             // the column's 32 tokens sit in the 4 lanes j, j+16, j+32, j+48
            // Deferred rescale: the running max is only advanced (and O / l rescaled) when some column's max grew by more than
            // 2^kDefer; until then p = exp2(s - m_run) may exceed 1 by that factor, harmless in fp16 / fp32.  In steady state
            // the per-group multiply of the 32 O accumulators (and their AGPR round trip) disappears.
            constexpr float kDefer = 6.f;
            const bool grow = __builtin_amdgcn_ballot_w64(mx > m_run[c] + kDefer) != 0;   // wave-uniform
            const float m_new = grow ? fmaxf(m_run[c], mx) : m_run[c];
            alpha[c] = grow ? __builtin_amdgcn_exp2f(m_run[c] - m_new) : 1.f;
            rescale |= grow;
            m_run[c] = m_new;
            float psum = 0.f;
            float pes[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool valid = !MASKED || vwin + e < limit[c];
                float pe = valid ? __builtin_amdgcn_exp2f(sv[e] - m_new) : 0.f;
                psum += pe;
                if (INT8) pe = valid ? pe * g.vsc[e >> 2][e & 3] : 0.f; // scale bytes past the context may be garbage
                pes[e] = pe;
            }
            pf[c] = act_pack8<BF>(pes);
            l_run[c] = l_run[c] * alpha[c] + psum;
            if (INT8) {
                const f16x2 ones = {(f16)1.f, (f16)1.f};
                const u32x4 pv = pf[c];
                float s16 = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) s16 = act_dot_ones<BF>(pv[e], s16);
                l16_run[c] = l16_run[c] * alpha[c] + s16;
            }
        }
        if (rescale) {
#pragma unroll
            for (int c = 0; c < NT; ++c)
#pragma unroll
                for (int db = 0; db < NDB; ++db) o[c][db] *= alpha[c];
        }
        // ---- O^T += V^T P
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            u32x4 a;
            if (INT8) {
                if constexpr (BF) a = widen_u8_bf16(g.vf8[db][0], g.vf8[db][1]);
                else              a = __builtin_bit_cast(u32x4, widen_kv8(g.vf8[db][0], g.vf8[db][1]));
            } else {
                a = g.vf16[db];
            }
            // tokens past the block's context carry p = 0 but V bytes there may be garbage (NaN/Inf): zero them
            if (MASKED && vwin + 7 >= seq_len) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (vwin + e >= seq_len) a[e >> 1] &= (e & 1) ? 0x0000FFFFu : 0xFFFF0000u;
            }
#pragma unroll
            for (int c = 0; c < NT; ++c) o[c][db] = mfma_act<BF>(a, pf[c], o[c][db]);
        }
    };

    // Two groups in flight per wave (registers ping-pong), block ids looked up one group further ahead:
    // with one long partition per (sequence, kv head) a SIMD holds a single wave, so the wave itself has to
    // keep >= 32 KB of K/V loads outstanding to cover HBM latency.
    {
        // a group is fully visible when it ends at or before the shortest context of the block's columns: no masking code
        int lim_min = limit[0];
#pragma unroll
        for (int c = 1; c < NT; ++c) lim_min = min(lim_min, limit[c]);
        lim_min = min(lim_min, __shfl_xor(lim_min, 1)); lim_min = min(lim_min, __shfl_xor(lim_min, 2));
        lim_min = min(lim_min, __shfl_xor(lim_min, 4)); lim_min = min(lim_min, __shfl_xor(lim_min, 8));
        const int full_end = __builtin_amdgcn_readfirstlane(p.R * p.G == 16 && nrows == NT * p.R ? lim_min : (p.q_len == 1 ? seq_len : 0));
        auto compute = [&](const Group& g, int tb) {
            if (tb + 32 <= full_end) compute_group(g, tb, std::false_type{});
            else compute_group(g, tb, std::true_type{});
        };
        // NG groups of K/V in flight per wave, the load of group n + NG - 1 issued BEFORE the wait for group n: a SIMD holds a
        // single wave here (one long partition per (sequence, kv head)), so the wave itself has to cover the HBM latency --
        // with a period of ~L / (NG - 1) per group instead of L (measured L ~ 3.5 us under load at b = 64).
        // (separate named groups, not an array: an indexed array of these structs ends up in scratch memory)
        Group g0, g1, g2;
        Wins w0, w1, w2;
        int tb = pstart + wave * 32;
        if constexpr (NG == 2) {
            if (tb < pend) { first_wins(w0); load_group(g0, tb, w0); }
            AT_STAMP(1);
            if (tb + GS < pend) lookup(tb + GS, w1);
            // Steady state without a single condition: both groups of the round and both groups it prefetches lie inside the
            // partition and need no mask.  (Guards around the loads make hipcc's waitcnt pass assume a load may have been
            // issued and never consumed: it then waits for the OTHER group's loads before it reuses a register.)
            while (tb + 3 * GS < pend && tb + GS + 32 <= full_end) {
                load_group(g1, tb + GS, w1);
                lookup(tb + 2 * GS, w0);
                compute_group(g0, tb, std::false_type{});
#ifdef MI355_TUNING
                if (tb == pstart + wave * 32) { asm volatile("s_nop 0" ::: "memory"); AT_STAMP(2); }
#endif
                load_group(g0, tb + 2 * GS, w0);
                lookup(tb + 3 * GS, w1);
                compute_group(g1, tb + GS, std::false_type{});
                tb += 2 * GS;
            }
            for (; tb < pend; tb += 2 * GS) {
                if (tb + GS < pend) load_group(g1, tb + GS, w1);
                if (tb + 2 * GS < pend) lookup(tb + 2 * GS, w0);
                compute(g0, tb);
                if (tb + 2 * GS < pend) load_group(g0, tb + 2 * GS, w0);
                if (tb + 3 * GS < pend) lookup(tb + 3 * GS, w1);
                if (tb + GS < pend) compute(g1, tb + GS);
            }
        } else if constexpr (NG == 4) {
            Group g3; Wins w3;
            if (tb < pend) { first_wins(w0); load_group(g0, tb, w0); }
            if (tb + GS < pend) { lookup(tb + GS, w1); load_group(g1, tb + GS, w1); }
            if (tb + 2 * GS < pend) { lookup(tb + 2 * GS, w2); load_group(g2, tb + 2 * GS, w2); }
            if (tb + 3 * GS < pend) lookup(tb + 3 * GS, w3);
            while (tb + 7 * GS < pend && tb + 3 * GS + 32 <= full_end) {
                load_group(g3, tb + 3 * GS, w3);
                lookup(tb + 4 * GS, w0);
                compute_group(g0, tb, std::false_type{});
                load_group(g0, tb + 4 * GS, w0);
                lookup(tb + 5 * GS, w1);
                compute_group(g1, tb + GS, std::false_type{});
                load_group(g1, tb + 5 * GS, w1);
                lookup(tb + 6 * GS, w2);
                compute_group(g2, tb + 2 * GS, std::false_type{});
                load_group(g2, tb + 6 * GS, w2);
                lookup(tb + 7 * GS, w3);
                compute_group(g3, tb + 3 * GS, std::false_type{});
                tb += 4 * GS;
            }
            for (; tb < pend; tb += 4 * GS) {
                if (tb + 3 * GS < pend) load_group(g3, tb + 3 * GS, w3);
                if (tb + 4 * GS < pend) lookup(tb + 4 * GS, w0);
                compute(g0, tb);
                if (tb + 4 * GS < pend) load_group(g0, tb + 4 * GS, w0);
                if (tb + 5 * GS < pend) lookup(tb + 5 * GS, w1);
                if (tb + GS < pend) compute(g1, tb + GS);
                if (tb + 5 * GS < pend) load_group(g1, tb + 5 * GS, w1);
                if (tb + 6 * GS < pend) lookup(tb + 6 * GS, w2);
                if (tb + 2 * GS < pend) compute(g2, tb + 2 * GS);
                if (tb + 6 * GS < pend) load_group(g2, tb + 6 * GS, w2);
                if (tb + 7 * GS < pend) lookup(tb + 7 * GS, w3);
                if (tb + 3 * GS < pend) compute(g3, tb + 3 * GS);
            }
        } else {
            if (tb < pend) { first_wins(w0); load_group(g0, tb, w0); }
            if (tb + GS < pend) { lookup(tb + GS, w1); load_group(g1, tb + GS, w1); }
            if (tb + 2 * GS < pend) lookup(tb + 2 * GS, w2);
            while (tb + 5 * GS < pend && tb + 2 * GS + 32 <= full_end) {
                load_group(g2, tb + 2 * GS, w2);
                lookup(tb + 3 * GS, w0);
                compute_group(g0, tb, std::false_type{});
                load_group(g0, tb + 3 * GS, w0);
                lookup(tb + 4 * GS, w1);
                compute_group(g1, tb + GS, std::false_type{});
                load_group(g1, tb + 4 * GS, w1);
                lookup(tb + 5 * GS, w2);
                compute_group(g2, tb + 2 * GS, std::false_type{});
                tb += 3 * GS;
            }
            for (; tb < pend; tb += 3 * GS) {
                if (tb + 2 * GS < pend) load_group(g2, tb + 2 * GS, w2);
                if (tb + 3 * GS < pend) lookup(tb + 3 * GS, w0);
                compute(g0, tb);
                if (tb + 3 * GS < pend) load_group(g0, tb + 3 * GS, w0);
                if (tb + 4 * GS < pend) lookup(tb + 4 * GS, w1);
                if (tb + GS < pend) compute(g1, tb + GS);
                if (tb + 4 * GS < pend) load_group(g1, tb + 4 * GS, w1);
                if (tb + 5 * GS < pend) lookup(tb + 5 * GS, w2);
                if (tb + 2 * GS < pend) compute(g2, tb + 2 * GS);
            }
        }
    }

    AT_STAMP(3);
    // ---- merge the NW waves through LDS, one column tile at a time.  o[c][db][r] is O^T[d = db*16 + w*4 + r][j].
    __shared__ float s_o[NW][16][HD + 4];
    __shared__ float s_m[NW][16], s_l[NW][16], s_b[NW][16];   // s_b: INT8 bias term 1152 * sum p16 of the wave's tokens
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        float lr = l_run[c];
        lr = xor32_sum(xor16_sum(lr));
        if (c) __syncthreads();
#pragma unroll
        for (int db = 0; db < NDB; ++db)
            *reinterpret_cast<f32x4*>(&s_o[wave][j][db * 16 + w * 4]) = o[c][db];
        float l16 = l16_run[c];
        if (INT8) l16 = xor32_sum(xor16_sum(l16));
        if (w == 0) { s_m[wave][j] = m_run[c]; s_l[wave][j] = lr; s_b[wave][j] = KVB * l16; }
        __syncthreads();
        AT_STAMP(4);
        // thread -> (column jj, 4 channels); 16 columns * HD/4 vectors
        const int ncol = p.R * p.G;
        for (int idx = tid; idx < ncol * (HD / 4); idx += NTHR) {
            const int jj = idx / (HD / 4), d0 = (idx - jj * (HD / 4)) * 4;
            const int rl = c * p.R + jj / p.G;
            if (rl >= nrows) continue;
            float mstar = s_m[0][jj];
#pragma unroll
            for (int ww = 1; ww < NW; ++ww) mstar = fmaxf(mstar, s_m[ww][jj]);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            float l = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) {
                const float f = __builtin_amdgcn_exp2f(s_m[ww][jj] - mstar);
                f32x4 ow = *reinterpret_cast<const f32x4*>(&s_o[ww][jj][d0]);
                if (INT8) ow -= s_b[ww][jj];
                acc += ow * f;
                l += s_l[ww][jj] * f;
            }
            const int row = row0 + rl, h = kh * p.G + (jj - (jj / p.G) * p.G);
            if (p.P == 1) {
                const float inv = l > 0.f ? 1.f / l : 0.f;          // padding row / empty context: zeros, not NaN
                u32x2 ov;
                if (BF && p.img_mblk > 0) {   // an image holds fp16 (common.h): the bf16-rounded outputs x 2^-8 (img_val)
                    ov[0] = act_pack<false>(img_val<true>(act_round<true>(acc[0] * inv)), img_val<true>(act_round<true>(acc[1] * inv)));
                    ov[1] = act_pack<false>(img_val<true>(act_round<true>(acc[2] * inv)), img_val<true>(act_round<true>(acc[3] * inv)));
                } else {
                    ov[0] = act_pack<BF>(acc[0] * inv, acc[1] * inv); ov[1] = act_pack<BF>(acc[2] * inv, acc[3] * inv);
                }
                *reinterpret_cast<u32x2*>(p.out + out_index(p, row, h * HD + d0, HD)) = ov;
            } else {
                const size_t slot = ((size_t)row * p.nh + h) * p.P + part;
                *reinterpret_cast<f32x4*>(p.tmp_out + slot * HD + d0) = acc;
                if (d0 == 0) { p.tmp_ml[slot * 2] = mstar; p.tmp_ml[slot * 2 + 1] = l; }
            }
        }
    }
    AT_STAMP(5);
}

// Merge partitions: one wave per (sequence, head); lane owns HD/64 channels.
template <int HD, bool BF = false>
__global__ __launch_bounds__(256) void attn_reduce_kernel(const AttnParams p) {
    const int lane = threadIdx.x & 63;
    const int gid = blockIdx.x * 4 + (threadIdx.x >> 6); // (row, h)
    if (gid >= p.B * p.q_len * p.nh) return;
    const int row = gid / p.nh;
    // ONE memory round trip: the row's length, the (max, sum) pairs (partition i in lane i) and the first 16 partial rows are requested
    // together; only the masks depend on the length.  (A loop bounded by the length waited for it, then walked the pairs one load at a
    // time: 5.6 us for this launch at one sequence.)  Partitions >= np were not written by this step: masked by select, never multiplied.
    const int sl = p.seq_lens[row];
    const float* __restrict__ ml = p.tmp_ml + (size_t)gid * p.P * 2;
    const float* __restrict__ po = p.tmp_out + (size_t)gid * p.P * HD + lane * (HD / 64);
    constexpr int CPL = HD / 64, UB = 16;
    float acc[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) acc[c] = 0.f;
    float l = 0.f, mstar = NEG_BIG;
    for (int i0 = 0; i0 < p.P; i0 += 64) {                                  // 64 partitions per pass (one pass up to 8192 tokens)
        float2 mlv = make_float2(NEG_BIG, 0.f);
        if (i0 + lane < p.P) mlv = *reinterpret_cast<const float2*>(ml + (size_t)(i0 + lane) * 2);
        float v[UB][CPL];
        const int nfirst = min(UB, p.P - i0);
#pragma unroll
        for (int u = 0; u < UB; ++u)
#pragma unroll
            for (int c = 0; c < CPL; ++c) v[u][c] = u < nfirst ? po[(size_t)(i0 + u) * HD + c] : 0.f;
        const int np = (min(max(sl + p.seq_add, 0), p.max_seq) + p.PS - 1) / p.PS;   // partitions that saw this row
        const bool mine = i0 + lane < np;
        const float m_i = mine ? mlv.x : NEG_BIG;
        const float mnew = fmaxf(mstar, wave_max(m_i));
        const float resc = __builtin_amdgcn_exp2f(mstar - mnew);            // earlier passes (1 on the first: NEG_BIG - NEG_BIG = 0)
        const float f_i = mine ? __builtin_amdgcn_exp2f(m_i - mnew) : 0.f;
        l *= resc;
#pragma unroll
        for (int c = 0; c < CPL; ++c) acc[c] *= resc;
        mstar = mnew;
        const int nhere = min(max(np - i0, 0), 64);
        auto lane_of = [&](float x, int i) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), i)); };
        // sums in partition order, as one thread walking the partitions would form them
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const bool on = u < nhere;
            const float f = lane_of(f_i, u), ly = on ? lane_of(mlv.y, u) : 0.f;
            l += ly * f;
#pragma unroll
            for (int c = 0; c < CPL; ++c) acc[c] += (on ? v[u][c] : 0.f) * f;
        }
        for (int i = UB; i < nhere; ++i) {                                   // long sequences at few rows: the rest
            const float f = lane_of(f_i, i);
            l += lane_of(mlv.y, i) * f;
#pragma unroll
            for (int c = 0; c < CPL; ++c) acc[c] += po[(size_t)(i0 + i) * HD + c] * f;
        }
    }
    const float inv = l > 0.f ? 1.f / l : 0.f;
    uint16_t* dst = reinterpret_cast<uint16_t*>(p.out) + out_index(p, row, (gid - row * p.nh) * HD + lane * CPL, HD);
#pragma unroll
    for (int c = 0; c < CPL; ++c) dst[c] = (BF && p.img_mblk > 0) ? act_to_bits<false>(img_val<true>(act_round<true>(acc[c] * inv))) : act_to_bits<BF>(acc[c] * inv);
}

#ifdef MI355_TUNING
int g_attn_ps_override = 0; // tuning hook (tools/attn_bench.py), tuning build only
#define ATTN_PS_OVERRIDE g_attn_ps_override
#else
#define ATTN_PS_OVERRIDE 0
#endif

int plan_partitions(int B, int nkv, int max_seq_len, int* ps_out) {
    if (ATTN_PS_OVERRIDE > 0) { *ps_out = ATTN_PS_OVERRIDE; return cdiv(max_seq_len, ATTN_PS_OVERRIDE); }
    // Long partitions stream best (measured b=64, ctx 1024: one 1024-token partition per (seq, kv head) reaches
    // 5.2 TB/s, five 256-token ones 4.0 TB/s and need the reduce kernel), and a CU holds ONE block of this kernel (> 256
    // registers per wave): split the sequence only while all blocks still fit one round of the 256 CUs, never below 128
    // tokens (one pass of the block's 4 waves).  Round 2 split until there were AT LEAST 256 blocks: 224 (sequence, kv head)
    // pairs became 448 half-length blocks = two rounds + the reduce launch, and b = 33..63 ran slower than b = 64.
    const long pairs = (long)B * nkv;
    int pmax = pairs >= 256 ? 1 : (int)(256 / pairs);
    int PS = cdiv(cdiv(max_seq_len, pmax), 128) * 128;
    if (PS < 128) PS = 128;
    *ps_out = PS;
    return cdiv(max_seq_len, PS);
}

} // namespace

#ifdef MI355_TUNING
extern "C" void mi355_debug_set_attn(int ps) { g_attn_ps_override = ps; }
static unsigned long long* g_attn_stamps = nullptr;
extern "C" void mi355_debug_attn_stamps(void* p) { g_attn_stamps = (unsigned long long*)p; }
#endif

extern "C" size_t mi355_paged_attn_workspace_bytes(int32_t B, int32_t nh, int32_t hd, int32_t max_seq_len) {
    if (B <= 0 || nh <= 0 || hd <= 0 || max_seq_len <= 0) return 0;
    const int P = cdiv(max_seq_len, 128); // upper bound over every plan
    return (size_t)B * nh * P * (hd + 2) * sizeof(float);
}

namespace {
int launch_attn(const void* q, const mi355_kv_layer_t* kv, const int32_t* block_table, int32_t max_blocks_per_seq,
                const int32_t* seq_lens, int32_t seq_lens_minus_one, int32_t B, int32_t q_len, int32_t nh, float scale,
                int32_t max_seq_len, void* out, void* workspace, size_t workspace_bytes, mi355_stream_t stream, int img_mblk = 0) {
    MI355_CHECK_ARG(q && kv && kv->kv_base && block_table && seq_lens && out, "paged_attn: null pointer");
    MI355_CHECK_ARG(kv->hd == 64 || kv->hd == 128, "paged_attn: hd=%d (64 or 128)", kv->hd);
    MI355_CHECK_ARG(kv->page >= 16 && kv->page % 8 == 0, "paged_attn: page=%d (>= 16, multiple of 8)", kv->page);
    MI355_CHECK_ARG(B > 0 && q_len > 0 && nh > 0 && kv->nkv > 0 && nh % kv->nkv == 0 && nh / kv->nkv <= 16,
                    "paged_attn: B=%d q_len=%d nh=%d nkv=%d (group <= 16)", B, q_len, nh, kv->nkv);
    MI355_CHECK_ARG(kv->num_blocks > 0, "paged_attn: num_blocks=%d", kv->num_blocks);
    MI355_CHECK_ARG(max_seq_len > 0 && (long)max_blocks_per_seq * kv->page >= max_seq_len,
                    "paged_attn: max_seq_len=%d exceeds block table", max_seq_len);
    MI355_CHECK_ARG((long)max_seq_len * kv->page < (1l << 32) && (long)kv->num_blocks * 2 * kv->nkv * kv->page < (1l << 31),
                    "paged_attn: max_seq_len * page >= 2^32 or scale plane >= 2^31 entries");
    const bool int8 = kv->kv_dtype == MI355_KV_INT8;
    MI355_CHECK_ARG(!int8 || kv->scale_base, "paged_attn: int8 cache needs scale_base");
    AttnParams p;
    p.q = (const f16*)q; p.kv_base = kv->kv_base; p.scale_base = kv->scale_base; p.block_table = block_table;
    p.seq_lens = seq_lens; p.out = (f16*)out; p.B = B; p.nh = nh; p.nkv = kv->nkv; p.G = nh / kv->nkv;
    p.page = kv->page; p.max_blocks = max_blocks_per_seq; p.seq_add = seq_lens_minus_one ? 1 : 0;
    p.max_seq = max_seq_len; p.num_blocks = kv->num_blocks;
    p.q_len = q_len; p.R = 16 / p.G; p.img_mblk = img_mblk;
#ifdef MI355_TUNING
    p.stamps = g_attn_stamps;
#endif
    p.page_magic = (uint32_t)((1ull << 32) / (unsigned)kv->page + 1);
    const int NT = q_len > p.R ? 2 : 1;                 // column tiles per block: 2 x R rows share one pass over the KV
    p.ntile = cdiv(q_len, NT * p.R);
    p.P = plan_partitions(B * p.ntile, kv->nkv, max_seq_len, &p.PS);
    p.scale_log2 = scale * 1.4426950408889634f;
    const size_t rows = (size_t)B * q_len;
    const size_t need = p.P > 1 ? rows * nh * p.P * (kv->hd + 2) * sizeof(float) : 0;
    if (need > workspace_bytes || (need && !workspace)) {
        mi355_set_error("paged_attn: workspace %zu < %zu", workspace_bytes, need);
        return MI355_ERR_WORKSPACE;
    }
    p.tmp_out = (float*)workspace;
    p.tmp_ml  = p.tmp_out + rows * nh * p.P * kv->hd;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(p.P, kv->nkv, B * p.ntile);
#define L_(HD_, I8_, NT_, NW_, NG_) hipLaunchKernelGGL((paged_attn_kernel<HD_, I8_, NT_, NW_, NG_>), grid, dim3(64 * NW_), 0, st, p)
#ifdef MI355_TUNING
#define L2_(HD_, I8_) do { if (NT == 1) { if (TUNE(6) == 3) L_(HD_, I8_, 1, 4, 3); else if (TUNE(6) == 4) L_(HD_, I8_, 1, 4, 4); else if (TUNE(6) == 2) L_(HD_, I8_, 1, 4, 2); \
                                            else if (TUNE(6) == 8) L_(HD_, I8_, 1, 8, 2); else if (TUNE(6) == 9) L_(HD_, I8_, 1, 8, 3); else L_(HD_, I8_, 1, 4, (I8_ ? 4 : 2)); } else L_(HD_, I8_, 2, 4, 2); } while (0)
#else
// INT8 groups are 8 KB: four of them in flight per wave (the bytes two fp16 groups hold); measured b = 64, ctx 4096: 77.2 -> 66.7 us
// (NG = 3: 72.7), ctx 1024 unchanged; fp16 loses with more than two (28.4 -> 30.6 us at ctx 1024)
#define L2_(HD_, I8_) do { if (NT == 1) L_(HD_, I8_, 1, 4, (I8_ ? 4 : 2)); else L_(HD_, I8_, 2, 4, 2); } while (0)
#endif
    const bool bf = kv->act_dtype == MI355_ACT_BF16;
    MI355_CHECK_ARG(kv->act_dtype == MI355_ACT_F16 || bf, "paged_attn: act_dtype=%d", kv->act_dtype);
    MI355_CHECK_ARG(int8 || (bf == (kv->kv_dtype == MI355_KV_BF16)), "paged_attn: a 16-bit cache has the dtype of the activations");
#define LB_(HD_, I8_) do { if (NT == 1) hipLaunchKernelGGL((paged_attn_kernel<HD_, I8_, 1, 4, (I8_ ? 4 : 2), true>), grid, dim3(256), 0, st, p); \
                           else         hipLaunchKernelGGL((paged_attn_kernel<HD_, I8_, 2, 4, 2, true>), grid, dim3(256), 0, st, p); } while (0)
    if (bf)            { if (kv->hd == 128) { if (int8) LB_(128, true); else LB_(128, false); } else { if (int8) LB_(64, true); else LB_(64, false); } }
    else if (kv->hd == 128) { if (int8) L2_(128, true); else L2_(128, false); }
    else               { if (int8) L2_(64, true);  else L2_(64, false); }
#undef LB_
#undef L2_
#undef L_
    MI355_CHECK_LAUNCH("paged_attn_kernel");
    if (p.P > 1) {
        const int nblk = cdiv((int)rows * nh, 4);
        if (bf) {
            if (kv->hd == 128) hipLaunchKernelGGL((attn_reduce_kernel<128, true>), dim3(nblk), dim3(256), 0, st, p);
            else               hipLaunchKernelGGL((attn_reduce_kernel<64, true>), dim3(nblk), dim3(256), 0, st, p);
        } else if (kv->hd == 128) hipLaunchKernelGGL(attn_reduce_kernel<128>, dim3(nblk), dim3(256), 0, st, p);
        else               hipLaunchKernelGGL(attn_reduce_kernel<64>, dim3(nblk), dim3(256), 0, st, p);
        MI355_CHECK_LAUNCH("attn_reduce_kernel");
    }
    return MI355_OK;
}
} // namespace

extern "C" int mi355_paged_decode_attn(const void* q, const mi355_kv_layer_t* kv, const int32_t* block_table,
                                       int32_t max_blocks_per_seq, const int32_t* seq_lens, int32_t B, int32_t nh,
                                       float scale, int32_t max_seq_len, void* out, void* workspace,
                                       size_t workspace_bytes, mi355_stream_t stream) {
    return launch_attn(q, kv, block_table, max_blocks_per_seq, seq_lens, 0, B, 1, nh, scale, max_seq_len, out, workspace,
                       workspace_bytes, stream);
}

extern "C" int mi355_paged_decode_attn_ex(const void* q, const mi355_kv_layer_t* kv, const int32_t* block_table,
                                          int32_t max_blocks_per_seq, const int32_t* seq_lens, int32_t seq_lens_minus_one,
                                          int32_t B, int32_t nh, float scale, int32_t max_seq_len, void* out,
                                          void* workspace, size_t workspace_bytes, mi355_stream_t stream) {
    return launch_attn(q, kv, block_table, max_blocks_per_seq, seq_lens, seq_lens_minus_one, B, 1, nh, scale, max_seq_len, out,
                       workspace, workspace_bytes, stream);
}

extern "C" int mi355_paged_attn_rows(const void* q, const mi355_kv_layer_t* kv, const int32_t* block_table,
                                     int32_t max_blocks_per_seq, const int32_t* positions, int32_t B, int32_t q_len, int32_t nh,
                                     float scale, int32_t max_seq_len, void* out, void* workspace, size_t workspace_bytes,
                                     mi355_stream_t stream) {
    return launch_attn(q, kv, block_table, max_blocks_per_seq, positions, 1, B, q_len, nh, scale, max_seq_len, out, workspace,
                       workspace_bytes, stream);
}

// the same with `out` written as an activation image (mi355_act_image_*: B * q_len <= 64 rows, K = nh * hd) for the O projection's
// full-K launch (mi355_linear_residual_img)
extern "C" int mi355_paged_attn_rows_img(const void* q, const mi355_kv_layer_t* kv, const int32_t* block_table,
                                         int32_t max_blocks_per_seq, const int32_t* positions, int32_t B, int32_t q_len, int32_t nh,
                                         float scale, int32_t max_seq_len, void* out_img, void* workspace, size_t workspace_bytes,
                                         mi355_stream_t stream) {
    MI355_CHECK_ARG(B > 0 && q_len > 0 && B * q_len <= 64 && kv && (nh * kv->hd) % 32 == 0, "paged_attn_rows_img: %d x %d rows (<= 64)", B, q_len);
    return launch_attn(q, kv, block_table, max_blocks_per_seq, positions, 1, B, q_len, nh, scale, max_seq_len, out_img, workspace,
                       workspace_bytes, stream, cdiv(B * q_len, 16));
}
