// Parameter block, epilogue kinds and the per-wave stamps shared by the full-K launches: gemm_fullk.hip (<= 16 rows and the
// generic shapes) and gemm_fullk64.hip (1-64 rows, W4 group-wise).
#pragma once
#include "gemm_common.h"

namespace {

struct RopeEpi {
    const float*   cos_sin;
    const int32_t* positions;
    const int32_t* block_table;
    int            max_blocks, nh, nkv, hd, page, max_pos, num_blocks, q_len;
    int32_t*       oob_count;
    void*          kv_base;
    f16*           q_out;
};
struct FullKParams {
    GemmParams g;
    const f16* res_in;
    f16*       res_out;
    RopeEpi    r;
    // NORM: x is the un-normed residual row h; the kernel applies RMSNorm on the fly, x_n = gamma * fp16(h * rs), with
    // rs = rsqrt(sum_k h^2 / K + eps) rebuilt from the per-tile partial sums the producing launch left in ssq_in
    const float* ssq_in;     // [rows][ssq_ld], ssq_ld >= ssq_tiles: sum over the 16 columns of tile t of h[row]^2
    int          ssq_tiles, ssq_ld;
    const f16*   gamma;
    float        eps;
    float*       ssq_out;    // FK_RESID: the same partial sums of the rows this launch produces ([M][ssq_ld]), or null
    int          ilv;        // K slices of the waves interleaved chunk by chunk (see the kernel)
    // FK_RESID of gemm_fullk64.hip, deferred RMSNorm of the rows it produces: besides h' it stores g = fp16(gamma 2^-e h') as an
    // activation image for the next GEMM, which multiplies its accumulators by rsqrt(mean h'^2 + eps) 2^e (from ssq_out)
    f16*         xg_img;     // image of g ([M][N]), or null
    const f16*   xg_gamma;   // norm weight [N]
    float        xg_scale;   // 2^-e with e >= log2(max |gamma|): |g| <= |h'|, no overflow whatever the residual stream holds
    int          rowsplit;   // gemm_fullk64.hip: two blocks per tile pair, 32 rows each (block b: pair b / 2, row blocks 2 (b & 1) .. + 1)
    int          bf16;       // gemm_fullk64.hip: bias / residual / norm weight / q / KV cache are bf16 (the activation image and the MFMAs stay fp16)
    // FK_PUB (gemm_fullk64.hip, tensor parallelism; round 6): y = 16-bit(xW + bias) goes straight into THIS RANK'S REGISTERED ALL-REDUCE BUFFER
    // (csrc/allreduce.hip), in the slot and parity the next fused all-reduce launch of that context reads -- a row-parallel shard (O / down) then needs
    // neither split-K slabs nor the fold + publish stage in front of the flag exchange (mi355_allreduce_fused_published_dt).  Row m of a <= 64-row
    // call lives in slot m (one row per block), parity (epoch[m] + 1) & 1 of the context's device-resident call counters.
    const uint32_t* pub_epoch;
    void*           pub_data;
    uint32_t        pub_bytes, pub_parity_elems, pub_slot_elems;
    int             pub_plain;  // 1: plain stores (full-fence hand-over: the all-reduce launch's system-scope release fence publishes them); 0: write-through (sc0 sc1)
#ifdef MI355_FULLK_STAMPS   // tuning build with MI355_EXTRA_CFLAGS=-DMI355_FULLK_STAMPS only: the stamp stores change the schedule
    unsigned long long* stamps;   // tools/fullk_stamps.py: wall_clock64 per wave at entry / requests out (+ 1 / rms there) / first chunk done / loop done / slices met / exit
#endif
};
#ifdef MI355_FULLK_STAMPS
#define FK_STAMP(i) do { if (fp.stamps && lane == 0) fp.stamps[((size_t)blockIdx.x * 16 + wave) * 6 + (i)] = wall_clock64(); } while (0)
#else
#define FK_STAMP(i) do { } while (0)
#endif
enum { FK_PLAIN = 0, FK_RESID = 1, FK_ROPE = 2, FK_PUB = 3 };

} // namespace
