// Full-K weight-only W4 GEMM for 1-64 rows (the step driver: from 5) with the consumer fused into the epilogue, gfx950: the QKV projection (+ bias + NeoX
// RoPE + Q extract + paged fp16 KV write) and the O projection (+ bias + residual add, per-tile sums of squares of the new
// residual rows) of a decode step as ONE launch each -- no split-K slabs, no fold launch.
// Reference semantics: LinearBase.forward (models_py/modules/factory/linear/linear_base.py:75-85) followed by
// FusedRopeKVCacheDecodeOp::forward (bindings/rocm/FusedRopeKVCacheOp.cc:519-646) resp. by the residual add of the decoder
// layer (model_desc/qwen3.py:63-77); module boundary modules/hybrid/causal_attention.py:75-93.
//
// What bounds it.  At these shapes (Qwen2-7B: K = 3584, N = 4608 / 3584) the weights are 7-9 MB -- 1.5 us of HBM time -- and the
// MFMA work is under a microsecond; what every block has to do is pull ALL 64 activation rows of its K range through its CU's
// vector-memory path, because K stays inside the block: 64 x 3584 x 2 B = 458 KB at the ~137 GB/s a CU takes out of L2
// (tools/probe/vmem_rate.hip: one 16-byte-per-lane wave-load per ~17 cycles) = 3.3 us, whatever the tile count of the block.  So:
//   * a block owns a PAIR of 16-column tiles (the (d, d + hd/2) pair of a head for the RoPE epilogue, adjacent tiles otherwise):
//     every activation fragment feeds two MFMAs, the grid is 144 / 112 blocks, and the XCDs' L2s serve 8 MB, not 13, each;
//   * its <= 16 waves are K slices of CPW chunks (128 k each).  A wave asks for ALL weights of its slice before anything else
//     (they come from HBM: longest latency, and vmcnt is an in-order queue -- a weight load issued between activation loads
//     would drain the activation ring when it is waited for), then streams the activation fragments through a register ring
//     of RING fragments in a fully static schedule: fragment f is consumed by TPB MFMAs and its slot re-requested at once, so
//     RING - 1 loads per wave stay in flight all the time (the full-K kernel of the few-row path kept half a chunk per wave
//     in flight and 7 waves per block: 20 us at 64 rows, profiles/r02_fullk_kernel_durations.txt);
//   * the slices meet once in LDS; wave mb sums row block mb of both tiles in slice order and runs the epilogue, whose
//     operands (bias / residual rows, position -> block id -> rotation row) were requested before the main loop.
// Weight image, dequant (operand side: exact subtract of the biased code, one rounding) and epilogue arithmetic are those of
// gemm_fullk.hip / rope_kv.hip, so the results are interchangeable with the composed launches.
// Round 5: per-channel W8 (load-time INT8 autoquant, device_impl.py:183-222) instances -- WB = 8: two wave-loads per (tile, chunk),
// the operand is the exact integer u - 128 (4 v_perm + 4 v_pk_add per 8 weights, no scale), the column's scale multiplies the
// summed accumulators in the epilogue (staged in LDS by the helper wave with the other epilogue operands).
#include "gemm_fullk.h"

namespace {

// LDS behind the slices' partial sums: what the epilogue needs besides the sums, staged by the helper wave
template <int EPI, int TPB> struct Stage64 {};
template <int TPB> struct Stage64<FK_RESID, TPB> { f16 res[64][16 * TPB]; f16 bias[16 * TPB]; f16 gam[16 * TPB]; float wsc[16 * TPB]; };   // residual rows / bias / norm weight of the block's tiles (+ W8: the columns' scales)
template <int TPB> struct Stage64<FK_PUB, TPB>   { f16 bias[16 * TPB]; float wsc[16 * TPB]; uint32_t par[64]; };                        // bias of the block's tiles, parity of every row's slot in the registered all-reduce buffer
template <int TPB> struct Stage64<FK_ROPE, TPB>  { float cs[64][16 * TPB]; int pos[64], blk[64]; f16 bias[16 * TPB]; float wsc[16 * TPB]; }; // rotation row of the block's 8 TPB dims per token, position, block id

// TPB (round 6): tiles per block.  2: one (d, d + hd/2) pair of a head / two adjacent tiles (rounds 4-5).  4: two pairs / four tiles with the rows split over MORE blocks
// -- what bounds these launches is the number of 1 KB wave-loads a block pushes through its CU's vector-memory path (tools/fullk64_stamps.py: the K waves' requests are out
// between 0.5 and 7 us of a 10.5 us launch), and per 8 MFMAs a (4 tiles x 2 row blocks) block loads 2 fragments + 1 weight KB where a (2 x 4) block loads 4 + 0.5:
// QKV at 64 rows 504 -> 336 wave-loads per CU on the same 144 blocks, O (4 tiles x 1 row block, 224 blocks) 280 -> 224.
template <int WB, int GS, int MB, int EPI, int CPW, int RING, int TPB = 2>
__global__ __launch_bounds__(1024) void gemm_fullk64_kernel(const FullKParams fp) {
    static_assert(TPB == 2 || TPB == 4, "tiles per block");
    constexpr bool W8 = WB == 8;                     // per-channel INT8 (GS is then a dummy 4)
    constexpr int LPC = WB / 4;                      // 1 KB wave-loads per (tile, chunk)
    constexpr int NSUB = 4 / GS, SPG = 4 / NSUB;     // GS: 4 -> g128, 2 -> g64, 1 -> g32: (zero, scale) words per chunk and column, k-steps per word
    constexpr int NF = CPW * 4 * MB;                 // activation fragments of a wave: (chunk c, k-step s, row block mb), f = (c * 4 + s) * MB + mb
    constexpr uint32_t FLAGS = 0x00020000u, OOBX = 0x80000000u, OOBS = 0x40000000u;   // out of range (lane / wave offsets)
    static_assert(RING <= NF && RING >= MB, "ring: at least one k-step, at most the slice");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* red = reinterpret_cast<f32x4*>(smem);     // [NW][TPB * MB][64]
    const GemmParams& p = fp.g;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NW   = (int)(blockDim.x >> 6) - 1;     // K-slice waves; wave NW is the helper
    const int jj = lane & 15, q = lane >> 4;
    Stage64<EPI, TPB>& sg = *reinterpret_cast<Stage64<EPI, TPB>*>(smem + (size_t)NW * TPB * MB * 1024);
    FK_STAMP(0);

    // rowsplit (O projection above 16 rows): the block's activation loads are what bounds the launch, and they are per ROW -- two
    // blocks per tile pair take MB row blocks each (16 rows up to 32, 32 above: 224 blocks for Qwen2-7B's O instead of 112; the weights
    // of a pair are then read twice, 56 KB more per CU pair).  Not for QKV: 144 pairs x 2 = 288 blocks, and the CUs with two of them would set the time.
    // The two blocks of a pair read the same weights: placed on the SAME XCD (block b runs on XCD b % 8 -- observed, used for speed
    // only) the second read comes out of that XCD's L2 instead of crossing the fabric again (rowsplit = 2: pairs a multiple of 8).
    // fp.rowsplit: 0 none | 1 two blocks per unit (b / 2, b % 2) | 2 the same with the two blocks of a unit on ONE XCD | 3 four blocks per unit | 4 four, on one XCD
    int unit = (int)blockIdx.x, part = 0;
    if (fp.rowsplit == 2)      { const int slot = (int)blockIdx.x >> 3; unit = (slot >> 1) * 8 + ((int)blockIdx.x & 7); part = slot & 1; }
    else if (fp.rowsplit == 1) { unit = (int)blockIdx.x >> 1; part = (int)blockIdx.x & 1; }
    else if (fp.rowsplit == 4) { const int slot = (int)blockIdx.x >> 3; unit = (slot >> 2) * 8 + ((int)blockIdx.x & 7); part = slot & 3; }
    else if (fp.rowsplit == 3) { unit = (int)blockIdx.x >> 2; part = (int)blockIdx.x & 3; }
    const int rb0 = part * MB;                       // first row block of this block (0 without a row split)
    int tile[TPB];
    if constexpr (EPI == FK_ROPE) {
        const int hh = fp.r.hd >> 5;                 // tiles per half head (even: the TPB / 2 pairs of a block belong to one head)
#pragma unroll
        for (int pp = 0; pp < TPB / 2; ++pp) {
            const int pair = unit * (TPB / 2) + pp, h = pair / hh, j = pair % hh;
            tile[2 * pp] = h * 2 * hh + j;
            tile[2 * pp + 1] = tile[2 * pp] + hh;
        }
    } else {
#pragma unroll
        for (int t = 0; t < TPB; ++t) tile[t] = unit * TPB + t;
    }

    if (wave == NW) {
        // =============================================================== helper wave: no K slice.  It stages the epilogue's operands
        // in LDS while the K waves stream: the chain position -> block id -> rotation row is two dependent round trips, and inside a
        // K wave every one of its waits would also wait for the weights and the ring in flight (vmcnt is in order)
        if constexpr (W8) {                          // per-channel scales of the block's 16 TPB columns: lane = (tile l / 16, column l % 16)
            float sc = 0.f;
            if (lane < 16 * TPB && tile[(lane >> 4) % TPB] < p.NT) sc = (float)as_h2(p.meta[tile[(lane >> 4) % TPB] * 16 + (lane & 15)])[1];
            if (lane < 16 * TPB) sg.wsc[lane] = sc;
        }
        constexpr int NPART = 2 * TPB;               // 16-byte parts of the block's 16 TPB columns
        if constexpr (EPI == FK_PUB) {
            // lane = row: the parity of its slot in the NEXT all-reduce call of the context (its blocks add 1 to epoch[row] when they finish)
            sg.par[lane] = (fp.pub_epoch[lane < p.M ? lane : 0] + 1u) & 1u;
            const int part_c = lane % NPART, n = tile[part_c >> 1] * 16 + (part_c & 1) * 8;
            u32x4 bv = {0u, 0u, 0u, 0u};
            if (p.bias && lane < NPART && n < p.N) bv = *reinterpret_cast<const u32x4*>(p.bias + n);
            if (lane < NPART) *reinterpret_cast<u32x4*>(&sg.bias[part_c * 8]) = bv;
        } else if constexpr (EPI == FK_RESID) {
            // this block's rows x 16 TPB columns of the residual stream: lane = (row l / NPART of the load, 16-byte part l % NPART); parts 2 t, 2 t + 1 = tile t
            __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)fp.res_in, 0, (uint32_t)((size_t)p.M * p.N * 2), FLAGS);
            constexpr int RPL = 64 / NPART, NL = MB * 16 / RPL;   // rows per load, loads for the block's MB row blocks
            const int part_c = lane % NPART, n = tile[part_c >> 1] * 16 + (part_c & 1) * 8;
            u32x4 rv[NL];
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int row = rb0 * 16 + i * RPL + lane / NPART;
                rv[i] = bload128<0>(rr, (row < p.M && n < p.N) ? (uint32_t)(((size_t)row * p.N + n) * 2) : OOBX);
            }
            u32x4 bv = {0u, 0u, 0u, 0u}, gv = bv;
            if (p.bias && lane < NPART && n < p.N) bv = *reinterpret_cast<const u32x4*>(p.bias + n);
            if (fp.xg_img && lane < NPART && n < p.N) gv = *reinterpret_cast<const u32x4*>(fp.xg_gamma + n);
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int row = rb0 * 16 + i * RPL + lane / NPART;
                if (row < 64) *reinterpret_cast<u32x4*>(&sg.res[row][part_c * 8]) = rv[i];
            }
            if (lane < NPART) { *reinterpret_cast<u32x4*>(&sg.bias[part_c * 8]) = bv; *reinterpret_cast<u32x4*>(&sg.gam[part_c * 8]) = gv; }
        } else {
            const RopeEpi& R = fp.r;
            const int half = R.hd >> 1, hh = R.hd >> 5;
            const int dj = (tile[0] % (2 * hh)) * 16;                                // first of the block's 8 TPB dims of the lower half (its pairs are adjacent: one run)
            const int row = lane < p.M ? lane : p.M - 1;                             // lane = token
            const int pos_in = R.positions[row];
            u32x4 bv = {0u, 0u, 0u, 0u};                                             // bias of the block's columns: lane = (tile l / 2, 16-byte part l % 2)
            if (p.bias && lane < NPART) bv = *reinterpret_cast<const u32x4*>(p.bias + tile[(lane >> 1) % TPB] * 16 + (lane & 1) * 8);
            const int pos = min(max(pos_in, 0), min(R.max_pos, R.max_blocks * R.page) - 1);
            const int blk = R.block_table[(size_t)(row / R.q_len) * R.max_blocks + pos / R.page];
            sg.pos[lane] = pos_in;
            if (lane < NPART) *reinterpret_cast<u32x4*>(&sg.bias[lane * 8]) = bv;
            // rotation rows of THIS block's tokens: a token's run is {cos, sin} x 8 TPB dims = 4 TPB 16-byte parts; load i covers tokens rb0 16 + i TPL + l / CPT, part l % CPT
            constexpr int CPT = 4 * TPB, TPL = 64 / CPT, NL = MB * 16 / TPL;
            f32x4 cv[NL];
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int tok = min(rb0 * 16 + i * TPL + lane / CPT, 63);
                const int pr = __shfl(pos, tok);
                cv[i] = *reinterpret_cast<const f32x4*>(R.cos_sin + ((size_t)pr * half + dj) * 2 + (lane % CPT) * 4);
            }
            sg.blk[lane] = blk;
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int tok = rb0 * 16 + i * TPL + lane / CPT;
                if (tok < 64) *reinterpret_cast<f32x4*>(&sg.cs[tok][(lane % CPT) * 4]) = cv[i];
            }
        }
        FK_STAMP(1);
    } else {
    // =================================================================== K-slice waves: chunks [c0, c0 + n_ch), n_ch <= CPW
    const uint32_t lane16 = lane * 16u, jj4 = jj * 4u;
    const int c0 = wave * CPW;
    const int n_ch = min(CPW, p.KC - c0);
    __amdgpu_buffer_rsrc_t rw[TPB], rm[TPB];
#pragma unroll
    for (int t = 0; t < TPB; ++t) {
        const bool ok = tile[t] < p.NT && !(fp.ilv & 2);
        const char* wb = (const char*)p.qw + ((size_t)tile[t] * p.KC + c0) * (LPC * 1024);
        rw[t] = __builtin_amdgcn_make_buffer_rsrc((void*)wb, 0, ok ? n_ch * LPC * 1024 : 0, FLAGS);
        const char* mb_ = (const char*)p.meta + ((size_t)c0 * NSUB * p.N_pad + tile[t] * 16) * 4;
        rm[t] = __builtin_amdgcn_make_buffer_rsrc((void*)mb_, 0, (ok && !W8) ? ((n_ch * NSUB - 1) * p.N_pad + 16) * 4 : 0, FLAGS);
    }
    // activations of the slice: the image's fragments (k-step 4 (c0 + c) + s, row block mb), 1 KB each (common.h act_img_index);
    // fp.ilv: timing experiments of the tuning build (1: no activation traffic, 2: no weight traffic -- same instruction stream)
    const int MBLK = (p.M + 15) >> 4;                // row blocks of the image (<= MB)
    const char* xb = (const char*)p.x + (size_t)c0 * 4 * MBLK * 1024;
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, (fp.ilv & 1) ? 0u : (uint32_t)(n_ch * 4 * MBLK * 1024), FLAGS);

    // ---- 1. all weights of the slice (HBM, non-temporal: read once by one CU)
    u32x4    wr[CPW][TPB][LPC];
    uint32_t mr[CPW][TPB][NSUB];
#pragma unroll
    for (int c = 0; c < CPW; ++c)
#pragma unroll
        for (int t = 0; t < TPB; ++t) {
#pragma unroll
            for (int lp = 0; lp < LPC; ++lp) wr[c][t][lp] = bload128<2 /*nt*/>(rw[t], lane16, (uint32_t)(c * LPC + lp) * 1024u);
            if constexpr (!W8) {
#pragma unroll
                for (int gi = 0; gi < NSUB; ++gi)
                    mr[c][t][gi] = __builtin_amdgcn_raw_buffer_load_b32(rm[t], jj4, (uint32_t)(c * NSUB + gi) * (uint32_t)p.N_pad * 4u, 0);
            }
        }

    // ---- 2. activation ring, fully static: slot f % RING holds fragment f
    u32x4 xr[RING];
    auto load_frag = [&](int f) {                    // f compile-time at every call site (static_for)
        const int c = f / (4 * MB), s = (f / MB) % 4, mb = f % MB;
        // a row block the image does not have (M <= 48 in the 64-row instance) or a chunk past the slice: past the descriptor, zeros
        xr[f % RING] = bload128<0>(rx, lane16, rb0 + mb < MBLK ? (uint32_t)(((c * 4 + s) * MBLK + rb0 + mb) * 1024) : OOBS);
    };
    static_for<0, RING>([&](auto f_) { load_frag(decltype(f_)::value); });
    FK_STAMP(1);

    f32x4 acc[TPB][MB];
#pragma unroll
    for (int t = 0; t < TPB; ++t)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[t][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const W4Consts w4c = w4_consts();
    const f16x2 c960 = {(f16)960.f, (f16)960.f};
    const f16x2 zn8 = {(f16)-1152.f, (f16)-1152.f};  // W8: byte u under the exponent of 1024 (v_perm), minus 1024 + 128: the exact integer q
    f16x8 a[TPB];
    static_for<0, NF>([&](auto f_) {
        constexpr int f = decltype(f_)::value, c = f / (4 * MB), s = (f / MB) % 4, mb = f % MB;
        if constexpr (mb == 0) {                     // A fragments of k-step (c, s): one dequant per tile feeds MB MFMAs
#pragma unroll
            for (int t = 0; t < TPB; ++t) {
                if constexpr (W8) {                  // wave-load s / 2 of the chunk holds k-steps 2 (s / 2), + 1: two dwords each
                    const u32x4 w = wr[c][t][s >> 1];
                    a[t] = dequant_w8<false>(w[(s & 1) * 2], w[(s & 1) * 2 + 1], zn8, zn8);
                } else {
                    const uint32_t m = mr[c][t][s / SPG];
                    const f16x2 zn = as_h2(__builtin_amdgcn_perm(m, m, 0x05040504u)), sc = as_h2(__builtin_amdgcn_perm(m, m, 0x07060706u));
                    a[t] = dequant_w4_vc(wr[c][t][0][s], zn, zn + c960, sc, w4c);
                }
            }
        }
        const f16x8 b = __builtin_bit_cast(f16x8, xr[f % RING]);
#pragma unroll
        for (int t = 0; t < TPB; ++t) acc[t][mb] = mfma16x16x32(a[t], b, acc[t][mb]);
        if constexpr (f + RING < NF) load_frag(f + RING);
        // fence per fragment: left free, hipcc sinks the ring's re-requests to just in front of their use (vmcnt(0..2) in the ISA:
        // one load in flight per wave instead of RING - 1)
        __builtin_amdgcn_sched_barrier(0);
#ifdef MI355_FULLK_STAMPS
        if constexpr (f == 0) { asm volatile("s_nop 0" ::: "memory"); FK_STAMP(2); }
#endif
    });
    FK_STAMP(3);

    // ---- 3. the K slices meet in LDS
#pragma unroll
    for (int t = 0; t < TPB; ++t)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) red[((size_t)wave * (TPB * MB) + t * MB + mb) * 64 + lane] = acc[t][mb];
    }   // K-slice waves
    __syncthreads();
    FK_STAMP(4);
    // Epilogue jobs: (tile, row block) for the plain epilogues, (tile pair of a head, row block) for RoPE; job j goes to wave j, j + NW + 1, ...: every job sums its
    // sets in slice order (the bits do not depend on which wave does it) and finishes them.  (Rounds 4-5 gave wave mb ALL tiles of row block mb: MB busy waves.)
    constexpr int TJ = (EPI == FK_ROPE) ? 2 : 1;    // tiles of a job
    constexpr int NJ = (TPB / TJ) * MB;
    for (int job = wave; job < NJ; job += NW + 1) {
    const int tg = job / MB, mb = job - tg * MB;     // tile group of the block, row block
    f32x4 v[TJ];
#pragma unroll
    for (int u = 0; u < TJ; ++u) {
        v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int w = 0; w < NW; ++w) v[u] += red[((size_t)w * (TPB * MB) + (tg * TJ + u) * MB + mb) * 64 + lane];
    }
    const int m = (rb0 + mb) * 16 + jj;
    if (m >= p.M) continue;

    const bool bf = fp.bf16 != 0;                   // dtype of everything 16-bit around the GEMM (the image and the weights' dequant are fp16)
    if constexpr (W8) {                             // per-channel scale of this lane's four columns
#pragma unroll
        for (int u = 0; u < TJ; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[u][r] *= sg.wsc[(tg * TJ + u) * 16 + q * 4 + r];
    }
    if (bf) {                                       // the image of a bf16 tensor holds x 2^-8 (common.h img_val): exact in fp32
#pragma unroll
        for (int u = 0; u < TJ; ++u) v[u] *= kImgBfUnscale;
    }
    if constexpr (EPI == FK_PUB) {
        const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(fp.pub_data, 0, fp.pub_bytes, FLAGS);
        const size_t rowbase = (size_t)sg.par[m] * fp.pub_parity_elems + (size_t)m * fp.pub_slot_elems;
        const int n0 = (unit * TPB + tg) * 16 + q * 4;
        if (n0 < p.N) {
            const uint16_t* bv = reinterpret_cast<const uint16_t*>(&sg.bias[tg * 16 + q * 4]);
            uint16_t ob[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) ob[r] = rt_to_bits(rt_round(v[0][r] + rt_from_bits(bv[r], bf), bf), bf);   // the linear's output is a 16-bit tensor
            const u32x2 o = {(uint32_t)ob[0] | ((uint32_t)ob[1] << 16), (uint32_t)ob[2] | ((uint32_t)ob[3] << 16)};
            if (fp.pub_plain) *reinterpret_cast<u32x2*>((f16*)fp.pub_data + rowbase + n0) = o;
            else __builtin_amdgcn_raw_buffer_store_b64(o, rp, (uint32_t)((rowbase + n0) * 2), 0, 17 /* sc0 | sc1: write-through, see allreduce.hip publish16 */);
        }
        FK_STAMP(5);
    } else if constexpr (EPI == FK_RESID) {
        const int tl = unit * TPB + tg, n0 = tl * 16 + q * 4;
        if (n0 < p.N) {
            const uint16_t* bv = reinterpret_cast<const uint16_t*>(&sg.bias[tg * 16 + q * 4]);
            const uint16_t* rin = reinterpret_cast<const uint16_t*>(&sg.res[m][tg * 16 + q * 4]);
            float of[4];
            uint16_t ob[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float y = rt_round(v[0][r] + rt_from_bits(bv[r], bf), bf);    // the linear's output is a 16-bit tensor in the reference
                of[r] = rt_round(y + rt_from_bits(rin[r], bf), bf);
                ob[r] = rt_to_bits(of[r], bf);
            }
            *reinterpret_cast<u32x2*>(fp.res_out + (size_t)m * p.N + n0) = (u32x2){(uint32_t)ob[0] | ((uint32_t)ob[1] << 16), (uint32_t)ob[2] | ((uint32_t)ob[3] << 16)};
            if (fp.xg_img) {                          // deferred RMSNorm: gamma 2^-e h' for the next GEMM (one rounding, from fp32; fp16 image)
                const uint16_t* gm = reinterpret_cast<const uint16_t*>(&sg.gam[tg * 16 + q * 4]);
                f16x4 g;
#pragma unroll
                for (int r = 0; r < 4; ++r) g[r] = (f16)(rt_from_bits(gm[r], bf) * fp.xg_scale * of[r]);
                *reinterpret_cast<f16x4*>(fp.xg_img + act_img_index(m, n0, (p.M + 15) >> 4)) = g;
            }
            if (fp.ssq_out) {                         // this tile's share of sum h'^2 of the row, for the consumer's RMSNorm
                float s2 = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) s2 += of[r] * of[r];
                s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
                if (q == 0) fp.ssq_out[(size_t)m * fp.ssq_ld + tl] = s2;
            }
        }
        FK_STAMP(5);
    } else if constexpr (EPI == FK_ROPE) {
        // tile pair tg of the block: this lane holds dims d0..d0+3 (v[0]) and d0+half..+3 (v[1]) of row m = token m
        const RopeEpi& R = fp.r;
        const int half = R.hd >> 1, hh = R.hd >> 5;
        const int pair = unit * (TPB / 2) + tg, tlo = (pair / hh) * 2 * hh + pair % hh;   // the pair's lower tile
        const int h  = tlo / (2 * hh);
        const int d0 = (tlo % (2 * hh)) * 16 + q * 4;
        float x0[4], x1[4];
        const uint16_t* b0 = reinterpret_cast<const uint16_t*>(&sg.bias[(2 * tg) * 16 + q * 4]);
        const uint16_t* b1 = reinterpret_cast<const uint16_t*>(&sg.bias[(2 * tg + 1) * 16 + q * 4]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            x0[r] = rt_round(v[0][r] + rt_from_bits(b0[r], bf), bf);
            x1[r] = rt_round(v[1][r] + rt_from_bits(b1[r], bf), bf);
        }
        const int pos_in  = sg.pos[m];
        const int pos_lim = min(R.max_pos, R.max_blocks * R.page);
        const int pos = min(max(pos_in, 0), pos_lim - 1);
        const bool is_v = h >= R.nh + R.nkv;
        if (!is_v) {
            const f32x4 cs01 = *reinterpret_cast<const f32x4*>(&sg.cs[m][tg * 32 + q * 8]), cs23 = *reinterpret_cast<const f32x4*>(&sg.cs[m][tg * 32 + q * 8 + 4]);
            const float cc[4] = {cs01[0], cs01[2], cs23[0], cs23[2]}, ss[4] = {cs01[1], cs01[3], cs23[1], cs23[3]};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float r0 = cc[r] * x0[r] - ss[r] * x1[r];
                const float r1 = cc[r] * x1[r] + ss[r] * x0[r];
                x0[r] = rt_round(r0, bf); x1[r] = rt_round(r1, bf);
            }
        }
        f16x4 o0, o1;                                 // 16-bit patterns of the activation dtype (moved as f16-typed words)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            o0[r] = __builtin_bit_cast(f16, rt_to_bits(x0[r], bf)); o1[r] = __builtin_bit_cast(f16, rt_to_bits(x1[r], bf));
        }
        FK_STAMP(5);
        if (h < R.nh) {
            f16* dst = R.q_out + ((size_t)m * R.nh + h) * R.hd + d0;
            *reinterpret_cast<f16x4*>(dst) = o0;
            *reinterpret_cast<f16x4*>(dst + half) = o1;
            continue;
        }
        const int kh = is_v ? h - R.nh - R.nkv : h - R.nh;
        if (pos_in < 0) continue;                                  // padding row of a multi-row step
        const int blk = sg.blk[m];
        if (pos != pos_in || blk < 0 || blk >= R.num_blocks) {   // stale position / block id: never write somebody else's page
            if (h == R.nh && d0 == 0 && R.oob_count) atomicAdd(R.oob_count, 1);
            continue;
        }
        const int tok = pos % R.page;
        const size_t head_elems = (size_t)R.page * R.hd;
        const size_t blk_base   = ((size_t)blk * 2 + (is_v ? 1 : 0)) * R.nkv + kh;
        f16* dst = (f16*)R.kv_base + blk_base * head_elems;
        if (!is_v) {
            *reinterpret_cast<f16x4*>(dst + tok * R.hd + d0) = o0;
            *reinterpret_cast<f16x4*>(dst + tok * R.hd + d0 + half) = o1;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) { dst[(d0 + r) * R.page + tok] = o0[r]; dst[(d0 + r + half) * R.page + tok] = o1[r]; }
        }
    }
    }   // epilogue jobs of this wave
    if constexpr (EPI == FK_PUB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's published bytes have reached memory before it ends (the kernel boundary then orders them in front of the all-reduce launch's flags)
}

template <int WB, int GS, int MB, int EPI, int CPW, int RING, int TPB = 2>
int launch64_t(const FullKParams& fp, int blocks, hipStream_t st) {
    auto k = gemm_fullk64_kernel<WB, GS, MB, EPI, CPW, RING, TPB>;
    const int NW = cdiv(fp.g.KC, CPW);
    if (NW > 15) return MI355_ERR_UNSUPPORTED;
    const size_t lds = (size_t)NW * TPB * MB * 1024 + sizeof(Stage64<EPI, TPB>);
    if (lds > 160 * 1024) return MI355_ERR_UNSUPPORTED;
    if (lds > 64 * 1024)
        if (int e = raise_dynamic_lds((const void*)k, "gemm_fullk64")) return e;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * (NW + 1)), lds, st, fp);
    MI355_CHECK_LAUNCH("gemm_fullk64_kernel");
    return MI355_OK;
}

// slice depth by K: up to 30 chunks (K <= 3840) two chunks per wave, up to 45 (K <= 5760) three; <= 15 K waves + the helper
template <int WB, int GS, int MB, int EPI>
int launch64_k(const FullKParams& fp, int blocks, hipStream_t st) {
    const int KC = fp.g.KC;
#ifdef MI355_TUNING
    if constexpr (WB == 4) {
    if (TUNE(6) == 1 && KC <= 30) return launch64_t<4, GS, MB, EPI, 2, MB>(fp, blocks, st);         // experiment: one k-step in flight
    if constexpr (GS == 4) if (TUNE(6) == 2 && KC <= 30) return launch64_t<4, GS, MB, EPI, 2, 3 * MB>(fp, blocks, st);     // experiment: three k-steps
    }
#endif
    if constexpr (WB == 8) {   // 32 bytes of weights per lane and chunk: the 64-row instance keeps one k-step of fragments in flight (128 registers)
        if (KC <= 30) return launch64_t<8, 4, MB, EPI, 2, MB == 4 ? MB : 2 * MB>(fp, blocks, st);
        return MI355_ERR_UNSUPPORTED;
    } else {
    // very short K (the O shard of a TP 8 rank: K = 1024, or TP 4 of Qwen2-7B: 896): one chunk per wave -- 7-8 K waves instead of 4, half the dependent fragment steps per wave.
    // Llama-3-70B tp 8 one rank's step 5.50 -> 5.47 ms; 14 chunks (Qwen2-7B tp 2: K = 1792) measured the same either way and keep two (profiles/r06_fullk64_short_k_one_chunk.txt)
    if (KC <= 8 && TUNE(14) != 1) return launch64_t<4, GS, MB, EPI, 1, 2 * MB>(fp, blocks, st);
    if (KC <= 30) return launch64_t<4, GS, MB, EPI, 2, 2 * MB>(fp, blocks, st);
    if (KC <= 45) return launch64_t<4, GS, MB, EPI, 3, (GS == 1 && MB == 4) ? MB : 2 * MB>(fp, blocks, st);   // g32 at 64 rows: 24 (zero, scale) words per wave, one k-step in flight fits 128 registers
    // K <= 9600 (hidden 8192: the QKV shard of Llama-3-70B / Qwen2-72B under TP 8): five chunks per wave; their 40 weight registers fit
    // the 128-register budget up to two row blocks per block -- the launcher only comes here with <= 32 rows per block (row split)
    if constexpr (MB <= 2 && GS == 4) { if (KC <= 75) return launch64_t<4, GS, MB, EPI, 5, 2 * MB>(fp, blocks, st); }
    return MI355_ERR_UNSUPPORTED;
    }
}

} // namespace

#ifdef MI355_FULLK_STAMPS
extern unsigned long long* g_fullk_stamps;   // gemm_fullk.hip
#endif

// fp_: a FullKParams filled by the entry points of gemm_fullk.hip (same layout in both translation units).  epi: FK_RESID or
// FK_ROPE.  Takes W4 group-wise weights, 1-64 rows, K <= 5760; MI355_ERR_UNSUPPORTED otherwise (the caller goes on to the
// generic full-K kernel).
extern "C" int mi355_gemm_fullk64(const void* fp_, int epi, int group_size, mi355_stream_t stream) {
    FullKParams fp = *reinterpret_cast<const FullKParams*>(fp_);
#ifdef MI355_FULLK_STAMPS
    fp.stamps = g_fullk_stamps;
#endif
    fp.ilv = TUNE(7); fp.rowsplit = 0;
    const GemmParams& g = fp.g;
    if (g.M < 1 || g.M > 64 || g.KC < 4 || g.KC > 75 || g.K != g.KC * 128) return MI355_ERR_UNSUPPORTED;
    const bool w8 = group_size == 0;                 // per-channel INT8 (the callers pass wbits == 8 as group_size 0)
    if (!w8 && group_size != 128 && group_size != 64 && group_size != 32) return MI355_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int mblk = (g.M + 15) >> 4;                // an instance per row-block count: the activation loads of a block are per row block
#define F64_(WB_, GS_, EPI_, BLOCKS_)                                                                        \
    return mblk == 1 ? launch64_k<WB_, GS_, 1, EPI_>(fp, BLOCKS_, st) : mblk == 2 ? launch64_k<WB_, GS_, 2, EPI_>(fp, BLOCKS_, st) \
         : mblk == 3 ? launch64_k<WB_, GS_, 3, EPI_>(fp, BLOCKS_, st) : launch64_k<WB_, GS_, 4, EPI_>(fp, BLOCKS_, st)
#define F64H_(WB_, GS_, EPI_) return mblk <= 2 ? launch64_k<WB_, GS_, 1, EPI_>(fp, 2 * blocks, st) : launch64_k<WB_, GS_, 2, EPI_>(fp, 2 * blocks, st)
#ifdef MI355_TUNING
    // round-6 experiment, tuning build only (switch 12 = 1): four tiles per block with the rows over more blocks (see the kernel): W4 g128, K <= 3840.  Measured SLOWER than
    // the pairs at 64 rows (QKV 11.5 vs 11.0 us, O 8.7 vs 8.0, profiles/r06_fullk64_quad_tiles_ab.txt): fewer wave-loads per CU do not shorten these launches.
    const bool quad_ok = !w8 && group_size == 128 && g.KC <= 30 && TUNE(12) == 1;
#endif
    if (epi == FK_ROPE) {
        if (fp.r.hd != 64 && fp.r.hd != 128) return MI355_ERR_UNSUPPORTED;
        const int blocks = (fp.r.nh + 2 * fp.r.nkv) * (fp.r.hd / 32);
#ifdef MI355_TUNING
        // 33-64 rows where two blocks per PAIR would not fit one round of the 256 CUs (Qwen2-7B: 144 pairs): two pairs of a head per block, two blocks of 32 rows per unit
        if (quad_ok && g.M > 32 && 2 * blocks > 256 && blocks % 2 == 0 && blocks <= 256) {
            fp.rowsplit = ((blocks / 2) % 8 == 0) ? 2 : 1;
            return launch64_t<4, 4, 2, FK_ROPE, 2, 4, 4>(fp, blocks, st);       // (blocks / 2 units) x 2 row halves
        }
#endif
        // a rank's shard of the QKV columns under tensor parallelism (Qwen2-7B tp2: 72 tile pairs, Llama-3-70B tp8: 40) or a small model
        // leaves more than half of the CUs without a block: two blocks per pair, half of the row blocks each, as for the O projection
        if (g.M > 16 && 2 * blocks <= 256 && !(TUNE(4) == 3)) {
            fp.rowsplit = (blocks % 8 == 0) ? 2 : 1;
            if (w8) { F64H_(8, 4, FK_ROPE); }
            if (group_size == 128) { F64H_(4, 4, FK_ROPE); }
            if (group_size == 64)  { F64H_(4, 2, FK_ROPE); }
            F64H_(4, 1, FK_ROPE);
        }
        if (w8) { F64_(8, 4, FK_ROPE, blocks); }
        if (group_size == 128) { F64_(4, 4, FK_ROPE, blocks); }
        if (group_size == 64)  { F64_(4, 2, FK_ROPE, blocks); }
        F64_(4, 1, FK_ROPE, blocks);
    }
    if (epi == FK_RESID) {
        const int blocks = cdiv(g.NT, 2);
#ifdef MI355_TUNING
        // 49-64 rows: four tiles x ONE row block per block, four blocks per unit (Qwen2-7B O: 56 units -> 224 blocks, as the two-way split of pairs)
        if (quad_ok && g.M > 48 && g.NT % 4 == 0 && g.NT <= 256) {
            fp.rowsplit = ((g.NT / 4) % 8 == 0) ? 4 : 3;
            return launch64_t<4, 4, 1, FK_RESID, 2, 2, 4>(fp, g.NT, st);
        }
#endif
        if (g.M > 16 && 2 * blocks <= 256 && !(TUNE(4) == 3)) {       // two blocks per tile pair, half of the row blocks each (see the kernel)
            fp.rowsplit = (blocks % 8 == 0) ? 2 : 1;
            if (w8) { F64H_(8, 4, FK_RESID); }
            if (group_size == 128) { F64H_(4, 4, FK_RESID); }
            if (group_size == 64)  { F64H_(4, 2, FK_RESID); }
            F64H_(4, 1, FK_RESID);
        }
        if (w8) { F64_(8, 4, FK_RESID, blocks); }
        if (group_size == 128) { F64_(4, 4, FK_RESID, blocks); }
        if (group_size == 64)  { F64_(4, 2, FK_RESID, blocks); }
        F64_(4, 1, FK_RESID, blocks);
    }
    if (epi == FK_PUB) {                             // a row-parallel TP shard straight into the registered all-reduce buffer: W4 g128 / per-channel W8
        if (!w8 && group_size != 128) return MI355_ERR_UNSUPPORTED;
        const int blocks = cdiv(g.NT, 2);
#ifdef MI355_TUNING
        if (quad_ok && g.M > 48 && g.NT % 4 == 0 && g.NT <= 256) {
            fp.rowsplit = ((g.NT / 4) % 8 == 0) ? 4 : 3;
            return launch64_t<4, 4, 1, FK_PUB, 2, 2, 4>(fp, g.NT, st);
        }
#endif
        if (g.M > 16 && 2 * blocks <= 256) {
            fp.rowsplit = (blocks % 8 == 0) ? 2 : 1;
            if (w8) { F64H_(8, 4, FK_PUB); }
            F64H_(4, 4, FK_PUB);
        }
        if (w8) { F64_(8, 4, FK_PUB, blocks); }
        F64_(4, 4, FK_PUB, blocks);
    }
#undef F64_
#undef F64H_
    return MI355_ERR_UNSUPPORTED;
}
