// One-shot all-reduce over peer-mapped memory (xGMI on a multi-GPU node), fused with the split-K reduce of the producing
// GEMM, the residual add and the RMSNorm of the consumer.  gfx950.
//
// Replaces the reference's custom all-reduce for the decode message sizes: TrtllmArFusionHandle
// (rtp_llm/models_py/bindings/rocm/TrtllmAllReduceFusion.h:14-55), kernels allreduce_kernel_1stage
// (bindings/rocm/kernels/trtllm_allreduce_fusion.cu:474-540), the fused all-reduce + residual + RMSNorm form
// allreduce_fusion_kernel_1stage (:431-471), the per-block flag barrier SyncComm (:285-316) and its numerics: peers are
// summed in fp32 in rank order 0..N-1 and rounded once, so every rank obtains bit-identical results (:228-246,488-506).
// Handle exchange mirrors base/rocm/trt_allreduce.py:51-230 (hipIpcGetMemHandle blobs gathered by the host).
//
// MI355X design.  8 GPUs are fully connected by point-to-point xGMI links, so for <= 1 MiB messages the cheapest
// all-reduce is "everybody reads everybody": each rank publishes its fp16 tensor in an IPC-exported buffer, raises a
// flag in every peer's flag table, waits for the peers' flags and then reads the N-1 remote copies concurrently over its
// 7 links while summing them.  One kernel per all-reduce point of the layer does ALL of:
//     stage 0   row <- sum of the local split-K slabs (+ bias), rounded to fp16      (what add_rmsnorm_kernel does at tp = 1)
//               published to this rank's registered buffer (double-buffered by the epoch's parity)
//     barrier   per block: block b only needs the rows block b of every peer published, so there is no grid-wide
//               synchronisation and no assumption about co-residency beyond "block b of every rank eventually runs"
//     stage 1   s = fp16(sum_r fp32(row_r)) in rank order; h = fp16(s + residual) -> residual stream; y = w * rmsnorm(h)
// Epochs live in device memory and advance inside the kernel, so a captured graph replays correctly; every spin is
// bounded (wall clock) and reports through a status word instead of hanging the GPU.
// Greedy sampling under a vocab-split lm_head uses the same transport for (max logit, global index) pairs instead of
// gathering the logits (PyWrappedModel.cc:915-936 gathers [B, V/tp] fp32 per rank; only 8 bytes per row are needed when
// every top_k == 1).
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <new>

#include "common.h"
#include "internal.h"

namespace {

constexpr int kMaxWorld  = 8;
constexpr int kMaxBlocks = 256;            // flag rows; grid of an all-reduce launch <= kMaxBlocks
constexpr int kFlagRow   = 16;             // dwords per block row: slots 0..7 first barrier, 8..15 second barrier (two-shot); one 64-byte line per block
constexpr unsigned long long kSpinTicks = 200000000ull;   // default bound: 2 s of the 100 MHz wall clock (mi355_allreduce_set_spin_timeout_ms)
constexpr int kOneShotRows = 64;          // tensors with more rows take the two-shot form (world > 2)
constexpr size_t kAuxBytes = 32768;        // per parity, after the tensor region: 8-byte records of the argmax exchange.
// Every location of a registered buffer has ONE owner block for all time, whatever the geometry (T, H) of the call: the tensor
// and result regions are cut into kMaxBlocks fixed SLOTS, block b keeps its rows b, b + grid, b + 2 grid, ... back to back
// inside slot b (row_off below), record r belongs to block r % kMaxBlocks.  A block alternates parities with its own epoch,
// so a location is rewritten two of its owner's calls after it was last read, and a peer can only have raised the flag of
// the call in between after finishing every earlier kernel on its stream.  (Round 2 laid rows out at r * H: the owner of a
// byte then changed with H, and back-to-back calls of different widths could overwrite rows a slow peer was still reading.)

struct ArDev {                             // device-visible part of the context (passed by value to kernels)
    f16*            my_data;               // registered buffer: 2 parities x max_elems
    const f16*      peer_data[kMaxWorld];
    uint32_t*       peer_flags[kMaxWorld]; // [kMaxBlocks][kFlagRow]; slot [b][r] is written by rank r
    uint32_t*       epoch;                 // [kMaxBlocks] this rank's call counter per block (ordinary device memory)
    int32_t*        status;                // != 0: a spin timed out (results invalid)
    size_t          parity_elems;          // elements between the two parities
    size_t          aux_elems;             // element offset of the record region inside a parity
    size_t          res_elems;             // element offset of the result region inside a parity (two-shot: rows this rank reduced)
    size_t          slot_elems;            // elements of one block's slot (tensor and result regions: kMaxBlocks slots each)
    uint32_t        data_bytes;            // bytes of one peer buffer (both parities): buffer range of the remote loads
    int             rank, world;
    unsigned long long spin_ticks;         // bound of every wait for a peer, in 100 MHz wall-clock ticks
    int             full_fences;           // 1: the round-1..4 hand-over (plain stores + system-scope release / acquire fences); 0: see publish16
    size_t          ll_elems;              // element offset of the granule region inside a parity (kMaxBlocks slots of 2 x slot_elems): see the LL form below
};

struct FusedParams {
    ArDev        ar;
    const f16*   x;            // fp16 [T][H] local tensor, or
    const float* partials;     // split-K slabs [nsplit][T][ld], or
    int          prepub;       // neither: the producing GEMM already wrote this rank's rows into the registered buffer (mi355_linear_publish_img): no stage 0
    int          nsplit, ld;
    const f16*   bias;         // added by rank 0 only (a row-parallel linear has one bias for the sum); may be null
    const f16*   res_in;       // residual stream in (may be null: plain all-reduce)
    f16*         res_out;      // residual stream out / plain all-reduce result
    const f16*   weight;       // RMSNorm weight (null: no norm)
    f16*         y;            // normed output
    int          y_img_mblk;   // > 0: y is an activation image of that many row blocks (common.h act_img_index: the next QKV launch reads it)
    float        eps;
    int          T, H;
    const uint32_t* pf;        // in-launch prefetch (mi355_allreduce_set_prefetch): the waves that only wait at the flag barrier touch
    uint32_t     pf_lines;     //   one dword of each of these 128-byte lines (null: off)
    uint32_t*    pf_sink;
};

// element offset (inside a region) of row `row` of a [T][H] tensor handled by a grid of `grid` blocks: slot row % grid, rows of
// one block back to back
__device__ __forceinline__ size_t row_off(const ArDev& ar, int row, int grid, int H) {
    return (size_t)(row % grid) * ar.slot_elems + (size_t)(row / grid) * (size_t)H;
}

// system-scope 16-byte load from a peer buffer (sc0 sc1: served by the owner's memory, never by a stale local line)
__device__ __forceinline__ u32x4 load_sys(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 17 /* sc0 | sc1 */);
}
// Publishing store into this rank's registered buffer.  Round 5: system-scope WRITE-THROUGH (sc0 sc1) -- the bytes go to memory, the
// line is not kept in the XCD's L2 -- so that what orders them in front of the flags is the wave's own `s_waitcnt vmcnt(0)` instead of
// a system-scope release fence.  That fence is `buffer_wbl2 sc0 sc1`: a write-back of the XCD's WHOLE L2 -- including the split-K
// slabs the GEMM in front just left there -- per block, and its acquire twin `buffer_inv sc0 sc1` drops every line the next GEMM
// could still have used; the peers' copies are read with sc0 sc1 loads, which never look at a local line, so nothing needs
// invalidating.  (LLVM's gfx942 memory model: a system-scope atomic store is `store sc0 sc1`, made visible in program order by
// `s_waitcnt vmcnt(0)`; MI355X_MICROARCH.md lists {sc0 sc1 stores and loads on both sides} + drained flag as a valid hand-over and
// prices the fence pair at 3.5 us and more once lines are dirty.)  One rank's step of tp = 2 (bench.py --shard-of 2): see
// profiles/r05_tp_allreduce_protocol.txt.  mi355_allreduce_set_full_fences(ar, 1) / MI355_AR_FULL_FENCES=1 bring the old form back.
__device__ __forceinline__ void publish16(const ArDev& ar, __amdgpu_buffer_rsrc_t mine, size_t elem_off, u32x4 v) {
    if (ar.full_fences) *reinterpret_cast<u32x4*>(ar.my_data + elem_off) = v;
    else __builtin_amdgcn_raw_buffer_store_b128(v, mine, (uint32_t)(elem_off * 2), 0, 17 /* sc0 | sc1 */);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t my_rsrc(const ArDev& ar) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)ar.my_data, 0, ar.data_bytes, 0x00020000u);
}

// Raise this block's flag at every peer, then wait for every peer's flag of the same epoch.  Executed by the whole block.
// slot0: 0 = the call's first barrier, 8 = the second barrier of a two-shot call (its own flag slots, same epoch value).
// pf: while thread t < world polls, waves 1.. of the block (which would only sit at the closing barrier) request the next GEMM's
// weight shard (mi355_allreduce_set_prefetch): kPfIters 128-byte lines per thread, issued AFTER the release fence so nothing is
// added in front of this block's flags; the words come back in pfv and are folded at the END of the kernel (pf_fold), never at
// the barrier.  vmcnt returns in order, so the peer reads of stage 1 queue behind these requests: the caller sizes the range.
constexpr int kPfIters = 8;
struct PfRegs { uint32_t v[kPfIters]; };
__device__ __forceinline__ void pf_fold(const PfRegs& r, uint32_t* sink) {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < kPfIters; ++i) acc ^= r.v[i];
    if (acc == 0x9E3779B9u && sink) *sink = acc;           // practically never: keeps the loads alive without a store per thread
}
__device__ __forceinline__ void peer_barrier(const ArDev& ar, int b, uint32_t epoch, int slot0 = 0, const uint32_t* pf = nullptr,
                                             uint32_t pf_lines = 0, PfRegs* pfv = nullptr) {
    if (ar.full_fences) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");          // system scope: this block's published rows first
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // every wave's write-through stores have reached memory (publish16)
    __syncthreads();
    const int t = threadIdx.x;
    if (pfv) {
        const uint32_t per = gridDim.x * (blockDim.x - 64);
        const uint32_t line = b * (blockDim.x - 64) + (t - 64);
#pragma unroll
        for (int i = 0; i < kPfIters; ++i)
            pfv->v[i] = (pf && t >= 64 && line + i * per < pf_lines) ? __builtin_nontemporal_load(pf + (size_t)(line + i * per) * 32) : 0u;
    }
    if (t < ar.world) {
        __hip_atomic_store(ar.peer_flags[t] + b * kFlagRow + slot0 + ar.rank, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const uint32_t* mine = ar.peer_flags[ar.rank] + b * kFlagRow + slot0 + t;
        const unsigned long long t0 = wall_clock64();
        // monotonic epochs: a peer may already be one call ahead (its next call's flag), never behind once it arrived
        while ((int32_t)(__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > ar.spin_ticks) { atomicExch(ar.status, 1 + t); break; }
        }
    }
    __syncthreads();
    if (ar.full_fences) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    else asm volatile("" ::: "memory");                    // the peers' rows are read with sc0 sc1 loads (load_sys): no local line to invalidate
}

// TWO = false: one-shot (every rank reads every peer's copy of every row: (N - 1) T H elements over its links -- cheapest for
// the <= 64 rows of a decode step).  TWO = true: two-shot for larger tensors (prefill chunks): rank r reduces the rows
// r, r + N, ... (rank-ordered fp32 sum of the N copies, one rounding -- the same numbers as the one-shot), publishes them in
// its result region, and after a second flag barrier every rank fetches each row ONCE from its owner: 2 (N - 1) / N T H
// elements per rank instead of (N - 1) T H (trtllm_allreduce_fusion.cu:606-692 is the reference's two-shot form).
// BF: the tensors (x, bias, residual, weight, y and the exchanged copies) are bf16; the sums stay fp32 in rank order
// LL = the low-latency form for the <= 64 rows of a decode step (one row per block): the published data carries its own flag.  Every 4 bytes of payload
// travel in a naturally aligned 8-byte GRANULE {payload, epoch} written by ONE write-through store (two granules per 16-byte store: a store may only ever
// tear between granules); a peer polls the granules themselves -- system-scope loads -- until every tag equals the call's epoch.  No flag table, no block
// barrier between publishing and pulling, the rank's own rows stay in registers: one fabric round trip per all-reduce instead of flag hop + data round trip
// (NCCL's LL protocol; MI355X_MICROARCH.md "handoff-1to1": data-tagged granules 0.8-1.0 us against 1.7-1.9 x that with a separate flag).  The granule region
// is double-buffered by the epoch's parity like the tensor region, tags are monotonic epochs (the buffer starts zeroed, epochs at 1), so a stale granule can
// never pass for a fresh one; the reuse argument of the tensor region holds unchanged (a rank can only publish call e + 2 after pulling every peer's call
// e + 1, which a peer publishes after it finished pulling call e).
template <int VPT, bool TWO, bool BF = false, bool LL = false>
__global__ __launch_bounds__(512) void allreduce_fused_kernel(const FusedParams p) {
    constexpr int NTH = 512;
    static_assert(!(LL && TWO), "the granule form is one-shot");
    const ArDev& ar = p.ar;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int nvec = p.H >> 3;
    const uint32_t epoch = ar.epoch[b] + 1;
    const size_t par = (epoch & 1) * ar.parity_elems;
    const __amdgpu_buffer_rsrc_t mine_rs = my_rsrc(ar);
    u32x4 own[VPT];                                    // LL: this rank's row (one row per block), kept for the rank-order sum
    // ---- stage 0: local row (split-K reduce + bias), fp16, into the registered buffer (prepub: the GEMM in front of this launch did that)
    for (int row = p.prepub ? p.T : b; row < p.T; row += gridDim.x) {
#pragma unroll
        for (int t = 0; t < VPT; ++t) {
            const int vi = tid + t * NTH;
            if (vi >= nvec) continue;
            const int c0 = vi * 8;
            u32x4 o;
            if (p.partials) {
                float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                const size_t sstride = (size_t)p.T * p.ld;
                const float* src0 = p.partials + (size_t)row * p.ld + c0;
                int s = 0;
                for (; s + 4 <= p.nsplit; s += 4) {            // 4 slabs per memory round trip, summed in index order
                    f32x4 a[4], c[4];                          // (as add_rmsnorm_kernel: same fp32 result, bit for bit)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        a[u] = *reinterpret_cast<const f32x4*>(src0 + (s + u) * sstride);
                        c[u] = *reinterpret_cast<const f32x4*>(src0 + (s + u) * sstride + 4);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[e] += a[u][e]; v[4 + e] += c[u][e]; }
                }
                for (; s < p.nsplit; ++s) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(src0 + s * sstride);
                    const f32x4 c = *reinterpret_cast<const f32x4*>(src0 + s * sstride + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += a[e]; v[4 + e] += c[e]; }
                }
                if (p.bias && ar.rank == 0) {
                    float bv[8];
                    act_unpack8<BF>(*reinterpret_cast<const u32x4*>(p.bias + c0), bv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += bv[e];
                }
                o = act_pack8<BF>(v);
            } else {
                o = *reinterpret_cast<const u32x4*>(p.x + (size_t)row * p.H + c0);
            }
            if constexpr (LL) {
                own[t] = o;
                const uint32_t goff = (uint32_t)((par + ar.ll_elems + (size_t)(row % gridDim.x) * 2 * ar.slot_elems + (size_t)(row / gridDim.x) * 2 * p.H + 2 * c0) * 2);
                __builtin_amdgcn_raw_buffer_store_b128((u32x4){o[0], epoch, o[1], epoch}, mine_rs, goff, 0, 17 /* sc0 | sc1 */);
                __builtin_amdgcn_raw_buffer_store_b128((u32x4){o[2], epoch, o[3], epoch}, mine_rs, goff + 16u, 0, 17);
            } else {
                publish16(ar, mine_rs, par + row_off(ar, row, gridDim.x, p.H) + c0, o);
            }
        }
    }
    PfRegs pfv;
    if constexpr (LL) {
#pragma unroll
        for (int i = 0; i < kPfIters; ++i) pfv.v[i] = 0u;
    } else {
        peer_barrier(ar, b, epoch, 0, p.pf, p.pf_lines, &pfv);
    }
    // ---- stage 1: rank-ordered fp32 sum of the N copies, residual add, RMSNorm
    __amdgpu_buffer_rsrc_t rp[kMaxWorld];
#pragma unroll
    for (int r = 0; r < kMaxWorld; ++r)
        rp[r] = __builtin_amdgcn_make_buffer_rsrc((void*)ar.peer_data[r < ar.world ? r : 0], 0, ar.data_bytes, 0x00020000u);
    if constexpr (TWO) {
        for (int row = b; row < p.T; row += gridDim.x) {
            if (row % ar.world != ar.rank) continue;                        // rows this rank reduces
#pragma unroll
            for (int t = 0; t < VPT; ++t) {
                const int vi = tid + t * NTH;
                if (vi >= nvec) continue;
                const uint32_t off = (uint32_t)((par + row_off(ar, row, gridDim.x, p.H) + vi * 8) * 2);
                u32x4 in[kMaxWorld];
#pragma unroll
                for (int r = 0; r < kMaxWorld; ++r)
                    if (r < ar.world) in[r] = load_sys(rp[r], off);
                float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < kMaxWorld; ++r) {
                    if (r < ar.world) {
                        float h[8];
                        act_unpack8<BF>(in[r], h);
#pragma unroll
                        for (int e = 0; e < 8; ++e) a[e] += h[e];
                    }
                }
                publish16(ar, mine_rs, par + ar.res_elems + row_off(ar, row, gridDim.x, p.H) + vi * 8, act_pack8<BF>(a));
            }
        }
        peer_barrier(ar, b, epoch, 8);
    }
    __shared__ float red[NTH / 64];
    for (int row = b; row < p.T; row += gridDim.x) {
        float v[VPT][8];
        u32x4 win[VPT];
        float ss = 0.f;
#pragma unroll
        for (int t = 0; t < VPT; ++t) {
            const int vi = tid + t * NTH;
            if (vi >= nvec) continue;
            const int c0 = vi * 8;
            const uint32_t off = (uint32_t)((par + row_off(ar, row, gridDim.x, p.H) + c0) * 2);
            u32x4 in[kMaxWorld];
            if constexpr (TWO) {                                            // the row as its owner reduced it
                const int owner = row % ar.world;
                const uint32_t roff = (uint32_t)((par + ar.res_elems + row_off(ar, row, gridDim.x, p.H) + c0) * 2);
#pragma unroll
                for (int r = 0; r < kMaxWorld; ++r)
                    if (r == owner) in[0] = load_sys(rp[r], roff);
            } else if constexpr (LL) {
                const uint32_t goff = (uint32_t)((par + ar.ll_elems + (size_t)(row % gridDim.x) * 2 * ar.slot_elems + (size_t)(row / gridDim.x) * 2 * p.H + 2 * c0) * 2);
                u32x4 ga[kMaxWorld], gb[kMaxWorld];
#pragma unroll
                for (int r = 0; r < kMaxWorld; ++r)
                    if (r < ar.world && r != ar.rank) { ga[r] = load_sys(rp[r], goff); gb[r] = load_sys(rp[r], goff + 16u); }   // all peers in flight together
                const unsigned long long t0 = wall_clock64();
#pragma unroll
                for (int r = 0; r < kMaxWorld; ++r) {
                    if (r < ar.world && r != ar.rank) {
                        while (ga[r][1] != epoch || ga[r][3] != epoch || gb[r][1] != epoch || gb[r][3] != epoch) {   // granules not there yet: ask again
                            if (wall_clock64() - t0 > ar.spin_ticks) { atomicExch(ar.status, 1 + r); break; }
                            __builtin_amdgcn_s_sleep(1);
                            ga[r] = load_sys(rp[r], goff); gb[r] = load_sys(rp[r], goff + 16u);
                        }
                        in[r] = (u32x4){ga[r][0], ga[r][2], gb[r][0], gb[r][2]};
                    } else if (r == ar.rank) {
                        in[r] = own[t];
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < kMaxWorld; ++r)
                    if (r < ar.world) in[r] = load_sys(rp[r], off);        // all peers in flight together
            }
            u32x4 rin = {0u, 0u, 0u, 0u};
            if (p.res_in) rin = *reinterpret_cast<const u32x4*>(p.res_in + (size_t)row * p.H + c0);
            win[t] = (p.y && p.weight) ? *reinterpret_cast<const u32x4*>(p.weight + c0) : (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
            for (int e = 0; e < 8; ++e) v[t][e] = 0.f;
            float h[8];
            if constexpr (TWO) {
                act_unpack8<BF>(in[0], h);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[t][e] = h[e];
            } else {
#pragma unroll
                for (int r = 0; r < kMaxWorld; ++r) {                       // rank order 0..N-1 on every rank
                    if (r < ar.world) {
                        act_unpack8<BF>(in[r], h);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[t][e] += h[e];
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[t][e] = act_round<BF>(v[t][e]);  // the all-reduced tensor is a 16-bit tensor: round once
            if (p.res_in) {
                act_unpack8<BF>(rin, h);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[t][e] = act_round<BF>(v[t][e] + h[e]);
            }
            if (p.res_out) *reinterpret_cast<u32x4*>(p.res_out + (size_t)row * p.H + c0) = act_pack8<BF>(v[t]);
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += v[t][e] * v[t][e];
        }
        if (!p.y) continue;
        ss = wave_sum(ss);                                                  // same reduction tree as add_rmsnorm_kernel
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = ss;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int i = 0; i < NTH / 64; ++i) tot += red[i];
        const float rs = rsqrtf(tot / (float)p.H + p.eps);
#pragma unroll
        for (int t = 0; t < VPT; ++t) {
            const int vi = tid + t * NTH;
            if (vi >= nvec) continue;
            float wv[8], o[8];
            act_unpack8<BF>(win[t], wv);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = wv[e] * act_round<BF>(v[t][e] * rs);
            if (p.y_img_mblk > 0) *reinterpret_cast<u32x4*>(p.y + act_img_index(row, vi * 8, p.y_img_mblk)) = img_pack8<BF>(o);   // the same 16 bytes at the image's address (fp16; bf16: x 2^-8)
            else                  *reinterpret_cast<u32x4*>(p.y + (size_t)row * p.H + vi * 8) = act_pack8<BF>(o);
        }
    }
    __syncthreads();
    if (tid == 0) ar.epoch[b] = epoch;
    pf_fold(pfv, p.pf_sink);
}

// All-gather along the hidden dimension over the same transport: every rank publishes its [T][n] column slice, then copies
// the N slices side by side -- out[t][r n + j] = slice_r[t][j], the result of the reference's hidden-split embedding
// (all_gather + reshape(tp, m, n).transpose(0, 1).reshape(m, -1), modules/base/common/embedding.py:50-58).  Rows keep the
// FULL row width n * world inside their block's slot (row_off), like the rows of an all-reduce: both kinds of call may
// alternate freely on one context.
struct GatherParams {
    ArDev      ar;
    const f16* x;      // [T][n]
    f16*       out;    // [T][n * world]
    int        T, n;
};

__global__ __launch_bounds__(256) void allgather_hidden_kernel(const GatherParams p) {
    const ArDev& ar = p.ar;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int H = p.n * ar.world, nv = p.n >> 3;
    const uint32_t epoch = ar.epoch[b] + 1;
    const size_t par = (epoch & 1) * ar.parity_elems;
    const __amdgpu_buffer_rsrc_t mine_rs = my_rsrc(ar);
    for (int row = b; row < p.T; row += gridDim.x)
        for (int vi = tid; vi < nv; vi += 256)
            publish16(ar, mine_rs, par + row_off(ar, row, gridDim.x, H) + vi * 8, *reinterpret_cast<const u32x4*>(p.x + (size_t)row * p.n + vi * 8));
    peer_barrier(ar, b, epoch);
#pragma unroll
    for (int r = 0; r < kMaxWorld; ++r) {
        if (r >= ar.world) break;
        __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)ar.peer_data[r], 0, ar.data_bytes, 0x00020000u);
        for (int row = b; row < p.T; row += gridDim.x)
            for (int vi = tid; vi < nv; vi += 256) {
                const u32x4 v = load_sys(rp, (uint32_t)((par + row_off(ar, row, gridDim.x, H) + vi * 8) * 2));
                *reinterpret_cast<u32x4*>(p.out + (size_t)row * H + (size_t)r * p.n + vi * 8) = v;
            }
    }
    __syncthreads();
    if (tid == 0) ar.epoch[b] = epoch;
}

// Greedy argmax across a vocab-split lm_head: every rank publishes (best local logit, global index) per row, then picks
// the overall best -- highest value, lowest global index on ties (torch.argmax semantics on the gathered row).
struct ArgmaxParams {
    ArDev        ar;
    const float* cand_v;       // [B][nparts] local candidates (argmax_stage1 of elementwise.hip)
    const int*   cand_i;
    int          nparts, B, vocab_offset;
    int32_t*     ids;
    int32_t*     positions;    // += 1 when non-null
};

__global__ __launch_bounds__(64) void allreduce_argmax_kernel(const ArgmaxParams p) {
    const ArDev& ar = p.ar;
    const int b = blockIdx.x, l = threadIdx.x;
    const uint32_t epoch = ar.epoch[b] + 1;
    const size_t par = (epoch & 1) * ar.parity_elems;
    for (int row = b; row < p.B; row += gridDim.x) {
        float bv = l < p.nparts ? p.cand_v[row * p.nparts + l] : -INFINITY;
        int   bi = l < p.nparts ? p.cand_i[row * p.nparts + l] : 0x7FFFFFFF;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (l == 0) {
            u32x2 rec = {__builtin_bit_cast(uint32_t, bv), (uint32_t)(bi + p.vocab_offset)};
            if (ar.full_fences) *reinterpret_cast<u32x2*>(ar.my_data + par + ar.aux_elems + (size_t)row * 4) = rec;   // 8 bytes per row
            else __builtin_amdgcn_raw_buffer_store_b64(rec, my_rsrc(ar), (uint32_t)((par + ar.aux_elems + (size_t)row * 4) * 2), 0, 17 /* sc0 | sc1: write-through, see publish16 */);
        }
    }
    peer_barrier(ar, b, epoch);
    for (int row = b; row < p.B; row += gridDim.x) {
        float bv = -INFINITY; int bi = 0x7FFFFFFF;
        if (l < ar.world) {
            __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)ar.peer_data[l], 0, ar.data_bytes, 0x00020000u);
            const u32x2 rec = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (uint32_t)((par + ar.aux_elems + (size_t)row * 4) * 2), 0, 17));
            bv = __builtin_bit_cast(float, rec[0]); bi = (int)rec[1];
        }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) {                                   // world <= 8 lanes
            const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (l == 0) {
            p.ids[row] = bi;
            if (p.positions) p.positions[row] += 1;
        }
    }
    __syncthreads();
    if (l == 0) ar.epoch[b] = epoch;
}

} // namespace

// ------------------------------------------------------------------ host side
struct mi355_allreduce {
    int     rank, world;
    size_t  max_bytes;                 // largest message (one parity)
    size_t  slot_bytes;                // one block's slot: >= its share of the largest message + one full-width row
    void*   data;                      // 2 parities x (tensor kMaxBlocks slots | records | results kMaxBlocks slots), IPC-exported
    void*   flags;                     // kMaxBlocks x kFlagRow dwords, IPC-exported
    void*   peer_data[kMaxWorld];
    void*   peer_flags[kMaxWorld];
    bool    opened[kMaxWorld];
    uint32_t* epoch;
    int32_t*  status;
    bool    ready;
    unsigned long long spin_ticks;
    const void* pf_ptr = nullptr;      // mi355_allreduce_set_prefetch: range the next fused launch touches while it waits
    size_t      pf_bytes = 0;
    int     full_fences = 0;           // mi355_allreduce_set_full_fences
    int     ll = 0;                    // mi355_allreduce_set_protocol(0) / MI355_AR_LL=1: the granule (LL) form for <= 64-row fused / sum calls (opt-in)
};

namespace {

struct HandleBlob {                    // what ranks exchange (host bytes)
    hipIpcMemHandle_t data, flags;
    int32_t           rank, world;
    uint64_t          max_bytes;
    int32_t           pid;
    int32_t           device;
};

// IPC-exportable device memory, zeroed.  Preferred: uncached (fine-grained) memory, whose stores are visible to peers
// without relying on a kernel boundary; if the runtime cannot export that kind, ordinary device memory (the kernels use
// system-scope fences and sc0 sc1 accesses on everything shared, so both kinds are correct).
void* alloc_shared(size_t bytes, hipIpcMemHandle_t* handle) {
    for (int attempt = 0; attempt < 2; ++attempt) {
        void* p = nullptr;
        const hipError_t e = attempt == 0 ? hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) : hipMalloc(&p, bytes);
        if (e == hipSuccess && p) {
            if (hipMemset(p, 0, bytes) == hipSuccess && hipDeviceSynchronize() == hipSuccess && hipIpcGetMemHandle(handle, p) == hipSuccess)
                return p;
            (void)hipFree(p);
        }
        (void)hipGetLastError();
    }
    return nullptr;
}

// Bytes of one parity of the registered buffer: tensor slots (kMaxBlocks) | records | two-shot result slots (kMaxBlocks) | granule slots.  The granule
// (LL) form runs one row per block on <= kOneShotRows blocks and a granule doubles its payload: kOneShotRows slots of 2 x slot_bytes (the
// round-5 layout reserved kMaxBlocks of them for an opt-in form and halved the largest message a context accepts).
static inline size_t parity_bytes(size_t slot_bytes) {
    return 2 * (size_t)kMaxBlocks * slot_bytes + kAuxBytes + 2 * (size_t)kOneShotRows * slot_bytes;
}

ArDev dev_view(const mi355_allreduce* a) {
    ArDev d;
    d.my_data = (f16*)a->data;
    for (int r = 0; r < kMaxWorld; ++r) {
        d.peer_data[r]  = (const f16*)a->peer_data[r < a->world ? r : 0];
        d.peer_flags[r] = (uint32_t*)a->peer_flags[r < a->world ? r : 0];
    }
    d.epoch = a->epoch; d.status = a->status;
    const size_t region = (size_t)kMaxBlocks * a->slot_bytes;
    d.parity_elems = parity_bytes(a->slot_bytes) / 2;       // per parity: tensor slots | records | two-shot result slots | granule slots (kOneShotRows x 2 slots)
    d.aux_elems = region / 2;
    d.res_elems = (region + kAuxBytes) / 2;
    d.ll_elems = (2 * region + kAuxBytes) / 2;
    d.slot_elems = a->slot_bytes / 2;
    d.data_bytes = (uint32_t)(2 * parity_bytes(a->slot_bytes));
    d.rank = a->rank; d.world = a->world; d.spin_ticks = a->spin_ticks; d.full_fences = a->full_fences;
    return d;
}

} // namespace

extern "C" size_t mi355_allreduce_handle_bytes(void) { return sizeof(HandleBlob); }

extern "C" mi355_allreduce_t* mi355_allreduce_create(int32_t rank, int32_t world, size_t max_bytes, void* handle_out) {
    // slot: the block's share of the largest message (rows of a block sit back to back) + one row of the widest tensor (8192 fp16)
    const size_t slot_bytes = ((max_bytes + kMaxBlocks - 1) / kMaxBlocks + 16384 + 255) & ~(size_t)255;
    if (rank < 0 || world < 1 || world > kMaxWorld || rank >= world || !handle_out || max_bytes == 0 ||
        2 * parity_bytes(slot_bytes) >= 0xFFFFFF00ull) {
        mi355_set_error("allreduce_create: rank=%d world=%d (1..%d) max_bytes=%zu", rank, world, kMaxWorld, max_bytes);
        return nullptr;
    }
    mi355_allreduce* a = new (std::nothrow) mi355_allreduce();
    if (!a) return nullptr;
    a->rank = rank; a->world = world; a->ready = false; a->spin_ticks = kSpinTicks;
    { const char* e = getenv("MI355_AR_FULL_FENCES"); a->full_fences = (e && e[0] == '1') ? 1 : 0; }
    { const char* e = getenv("MI355_AR_LL"); a->ll = (!a->full_fences && e && e[0] == '1') ? 1 : 0; }
    a->max_bytes = (max_bytes + 255) & ~(size_t)255;
    a->slot_bytes = slot_bytes;
    for (int r = 0; r < kMaxWorld; ++r) { a->peer_data[r] = a->peer_flags[r] = nullptr; a->opened[r] = false; }
    HandleBlob hb;
    memset(&hb, 0, sizeof(hb));
    a->data  = alloc_shared(2 * parity_bytes(slot_bytes), &hb.data);   // per parity: tensor slots | records | two-shot result slots | granule slots
    a->flags = alloc_shared((size_t)kMaxBlocks * kFlagRow * 4, &hb.flags);
    a->epoch = nullptr; a->status = nullptr;
    if (!a->data || !a->flags || hipMalloc((void**)&a->epoch, kMaxBlocks * 4 + 256) != hipSuccess ||
        hipMemset(a->epoch, 0, kMaxBlocks * 4 + 256) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        mi355_set_error("allreduce_create: cannot allocate / export the shared buffers: %s", hipGetErrorString(hipGetLastError()));
        mi355_allreduce_destroy(a);
        return nullptr;
    }
    a->status = (int32_t*)(a->epoch + kMaxBlocks);
    hb.rank = rank; hb.world = world; hb.max_bytes = a->max_bytes; hb.pid = (int32_t)getpid();
    (void)hipGetDevice(&hb.device);
    memcpy(handle_out, &hb, sizeof(hb));
    a->peer_data[rank] = a->data; a->peer_flags[rank] = a->flags;
    return a;
}

extern "C" int mi355_allreduce_open(mi355_allreduce_t* a, const void* all_handles) {
    MI355_CHECK_ARG(a && all_handles, "allreduce_open: null argument");
    const HandleBlob* hb = (const HandleBlob*)all_handles;
    for (int r = 0; r < a->world; ++r) {
        MI355_CHECK_ARG(hb[r].rank == r && hb[r].world == a->world && hb[r].max_bytes == a->max_bytes,
                        "allreduce_open: handle %d is from rank %d / world %d / %llu bytes", r, hb[r].rank, hb[r].world,
                        (unsigned long long)hb[r].max_bytes);
        if (r == a->rank) continue;
        if (hipIpcOpenMemHandle(&a->peer_data[r], hb[r].data, hipIpcMemLazyEnablePeerAccess) != hipSuccess ||
            hipIpcOpenMemHandle(&a->peer_flags[r], hb[r].flags, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
            mi355_set_error("allreduce_open: hipIpcOpenMemHandle(rank %d): %s", r, hipGetErrorString(hipGetLastError()));
            return MI355_ERR_HIP;
        }
        a->opened[r] = true;
    }
    a->ready = true;
    return MI355_OK;
}

extern "C" void mi355_allreduce_destroy(mi355_allreduce_t* a) {
    if (!a) return;
    for (int r = 0; r < kMaxWorld; ++r) {
        if (a->opened[r]) { if (a->peer_data[r]) hipIpcCloseMemHandle(a->peer_data[r]); if (a->peer_flags[r]) hipIpcCloseMemHandle(a->peer_flags[r]); }
    }
    if (a->data) hipFree(a->data);
    if (a->flags) hipFree(a->flags);
    if (a->epoch) hipFree(a->epoch);
    (void)hipGetLastError();
    delete a;
}

// Bound of every in-kernel wait for a peer (default 2 s).  One process per GPU never comes near it; ranks that SHARE a device (the
// single-GPU validation of the collectives, several test workers on one box) are time-sliced against each other and against whatever
// else runs there, so their contexts ask for a longer bound.  Takes effect for launches enqueued (or captured) afterwards.
extern "C" int mi355_allreduce_set_spin_timeout_ms(mi355_allreduce_t* a, int32_t ms) {
    MI355_CHECK_ARG(a && ms >= 1 && ms <= 600000, "allreduce_set_spin_timeout_ms: ms=%d (1..600000)", ms);
    a->spin_ticks = (unsigned long long)ms * 100000ull;
    return MI355_OK;
}

// Hand-over protocol of every later launch of this context: 0 (default) = write-through publishing stores + drained flags (publish16),
// 1 = plain stores between system-scope release / acquire fences (rounds 1-4).  Same results either way; every rank of a group must
// use the same setting only for timing comparisons -- the two forms interoperate.
extern "C" int mi355_allreduce_set_full_fences(mi355_allreduce_t* a, int32_t on) {
    MI355_CHECK_ARG(a, "allreduce_set_full_fences: null context");
    a->full_fences = on ? 1 : 0;
    a->ll = 0;                                   // either value names a flag form: the granule form is only reachable through mi355_allreduce_set_protocol(a, 0)
    return MI355_OK;
}

// 1 (default): write-through publishing stores + flags everywhere; 0: data-tagged granules (LL) for the <= 64-row one-shot calls + write-through stores for
// the rest -- OPT-IN: on one GPU it saves ~1 % of a rank's step (profiles/r05_tp_allreduce_granules.txt; the hop it removes is an xGMI hop), passes the
// two- and four-process checks, and times out with EIGHT processes time-slicing one GPU (every thread of every block polls: the single-GPU setup cannot say
// whether a real node likes it); 2: plain stores between system-scope release / acquire fences (rounds 1-4).  Same results in all three.
extern "C" int mi355_allreduce_set_protocol(mi355_allreduce_t* a, int32_t mode) {
    MI355_CHECK_ARG(a && mode >= 0 && mode <= 2, "allreduce_set_protocol: mode=%d (0..2)", mode);
    a->ll = mode == 0; a->full_fences = mode == 2;
    return MI355_OK;
}

// In-launch prefetch for the NEXT mi355_allreduce_fused[_dt] launch of this context (then cleared): while a block waits for its
// peers' flags, its otherwise idle waves touch one dword per 128-byte line of [ptr, ptr + bytes) -- the shard of the GEMM that
// consumes the all-reduce -- so the fetch from HBM runs under the xGMI exchange without a second stream or a graph edge
// (profiles/r03_prefetch_sidestream_ab.txt: every fork / join pair costs ~17 us of a step).  A launch of `grid` blocks covers at most
// grid * 448 * 8 lines (28 MB at 64 rows); the rest of the range is simply not touched.  ptr == NULL or bytes == 0: off.
extern "C" int mi355_allreduce_set_prefetch(mi355_allreduce_t* a, const void* ptr, size_t bytes) {
    MI355_CHECK_ARG(a && ((uintptr_t)ptr & 3) == 0, "allreduce_set_prefetch: null context or unaligned pointer");
    a->pf_ptr = bytes ? ptr : nullptr;
    a->pf_bytes = ptr ? (bytes < ((size_t)1 << 38) ? bytes : ((size_t)1 << 38)) : 0;
    return MI355_OK;
}

// Clear the status word (a bounded spin that timed out leaves it set for good: the host decides when a context is usable again)
extern "C" int mi355_allreduce_clear_status(mi355_allreduce_t* a, mi355_stream_t stream) {
    MI355_CHECK_ARG(a, "allreduce_clear_status: null");
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess || hipMemset(a->status, 0, 4) != hipSuccess) {
        mi355_set_error("allreduce_clear_status: %s", hipGetErrorString(hipGetLastError()));
        return MI355_ERR_HIP;
    }
    return MI355_OK;
}

extern "C" int mi355_allreduce_status(mi355_allreduce_t* a, mi355_stream_t stream) {
    MI355_CHECK_ARG(a, "allreduce_status: null");
    int32_t v = 0;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess || hipMemcpy(&v, a->status, 4, hipMemcpyDeviceToHost) != hipSuccess) {
        mi355_set_error("allreduce_status: %s", hipGetErrorString(hipGetLastError()));
        return MI355_ERR_HIP;
    }
    return v;
}

extern "C" int mi355_allreduce_fused(mi355_allreduce_t* a, const void* x_f16, const float* partials, int32_t nsplit, int32_t ld,
                                     const void* bias, const void* residual_in, void* residual_out, const void* weight, float eps,
                                     int32_t T, int32_t H, void* y, mi355_stream_t stream) {
    return mi355_allreduce_fused_dt(a, x_f16, partials, nsplit, ld, bias, residual_in, residual_out, weight, eps, T, H, y, MI355_ACT_F16, stream);
}

static int allreduce_fused_launch(mi355_allreduce_t* a, const void* x_f16, const float* partials, int32_t nsplit, int32_t ld,
                                  const void* bias, const void* residual_in, void* residual_out, const void* weight, float eps,
                                  int32_t T, int32_t H, void* y, int y_img_mblk, int32_t act_dtype, mi355_stream_t stream, bool prepub = false);

extern "C" int mi355_allreduce_fused_dt(mi355_allreduce_t* a, const void* x_f16, const float* partials, int32_t nsplit, int32_t ld,
                                        const void* bias, const void* residual_in, void* residual_out, const void* weight, float eps,
                                        int32_t T, int32_t H, void* y, int32_t act_dtype, mi355_stream_t stream) {
    return allreduce_fused_launch(a, x_f16, partials, nsplit, ld, bias, residual_in, residual_out, weight, eps, T, H, y, 0, act_dtype, stream);
}

// the same with the normed output written as an activation image (mi355_act_image_*, <= 64 rows): what the next layer's QKV launch on
// images reads under tensor parallelism (the tp = 1 step gets it from mi355_add_rmsnorm_img)
extern "C" int mi355_allreduce_fused_img_dt(mi355_allreduce_t* a, const void* x_f16, const float* partials, int32_t nsplit, int32_t ld,
                                            const void* bias, const void* residual_in, void* residual_out, const void* weight, float eps,
                                            int32_t T, int32_t H, void* y_img, int32_t act_dtype, mi355_stream_t stream) {
    MI355_CHECK_ARG(y_img && T <= 64 && H % 32 == 0, "allreduce_fused_img: T=%d (<= 64) H=%d (%% 32 == 0), y_img required", T, H);
    return allreduce_fused_launch(a, x_f16, partials, nsplit, ld, bias, residual_in, residual_out, weight, eps, T, H, y_img, cdiv(T, 16), act_dtype, stream);
}

// Where a producing GEMM writes this rank's [T][H] rows so that the NEXT fused all-reduce launch of the context finds them published
// (gemm_fullk64.hip FK_PUB via mi355_linear_publish_img; consumer mi355_allreduce_fused_published_dt).  T <= 64: one row per block, row m in slot m of
// the parity (epoch[m] + 1) & 1, which the GEMM reads off the device-resident counters -- inside a captured graph too.  Not with the granule form
// (its rows travel as {payload, epoch} pairs written by the all-reduce launch itself).  MI355_ERR_UNSUPPORTED: the caller keeps the slab path.
extern "C" int mi355_allreduce_publish_target(mi355_allreduce_t* a, int32_t T, int32_t H, mi355_publish_target_t* out) {
    MI355_CHECK_ARG(a && out, "allreduce_publish_target: null argument");
    if (!a->ready || a->ll || T < 1 || T > kOneShotRows || H < 8 || H % 8 != 0 || H > 8192 || (size_t)H * 2 > a->slot_bytes || (size_t)T * H * 2 > a->max_bytes)
        return MI355_ERR_UNSUPPORTED;
    const ArDev d = dev_view(a);
    out->epoch = a->epoch; out->data = a->data; out->bytes = d.data_bytes;
    out->parity_elems = (uint32_t)d.parity_elems; out->slot_elems = (uint32_t)d.slot_elems; out->plain_stores = a->full_fences;
    return MI355_OK;
}

// mi355_allreduce_fused[_img]_dt for rows the GEMM in front of it already published (mi355_linear_publish_img on the same stream, no other call of this
// context in between): flag exchange, rank-order fp32 sum of the N copies, residual add, RMSNorm -- no slab fold, no publishing stage.
extern "C" int mi355_allreduce_fused_published_dt(mi355_allreduce_t* a, const void* residual_in, void* residual_out, const void* weight, float eps,
                                                  int32_t T, int32_t H, void* y, int32_t y_is_image, int32_t act_dtype, mi355_stream_t stream) {
    MI355_CHECK_ARG(!y_is_image || (y && T <= 64 && H % 32 == 0), "allreduce_fused_published: image output needs y, T <= 64, H %% 32 == 0 (T=%d H=%d)", T, H);
    return allreduce_fused_launch(a, nullptr, nullptr, 0, 0, nullptr, residual_in, residual_out, weight, eps, T, H, y, y_is_image ? cdiv(T, 16) : 0, act_dtype, stream, true);
}

static int allreduce_fused_launch(mi355_allreduce_t* a, const void* x_f16, const float* partials, int32_t nsplit, int32_t ld,
                                  const void* bias, const void* residual_in, void* residual_out, const void* weight, float eps,
                                  int32_t T, int32_t H, void* y, int y_img_mblk, int32_t act_dtype, mi355_stream_t stream, bool prepub) {
    MI355_CHECK_ARG(a && a->ready, "allreduce: context not opened (mi355_allreduce_open)");
    MI355_CHECK_ARG(act_dtype == MI355_ACT_F16 || act_dtype == MI355_ACT_BF16, "allreduce: act_dtype=%d", act_dtype);
    MI355_CHECK_ARG(prepub ? (!x_f16 && !partials && T <= kOneShotRows && !a->ll) : ((x_f16 != nullptr) != (partials != nullptr)),
                    "allreduce: exactly one of x_f16 / partials (published rows: neither, T <= %d, not the granule form)", kOneShotRows);
    MI355_CHECK_ARG(T > 0 && H > 0 && H % 8 == 0 && H <= 8192, "allreduce: T=%d H=%d (H %% 8 == 0, H <= 8192)", T, H);
    MI355_CHECK_ARG((size_t)T * H * 2 <= a->max_bytes, "allreduce: message %zu bytes > registered %zu", (size_t)T * H * 2, a->max_bytes);
    MI355_CHECK_ARG(!partials || (nsplit >= 1 && ld >= H && ld % 4 == 0), "allreduce: nsplit=%d ld=%d", nsplit, ld);
    MI355_CHECK_ARG(residual_out || y, "allreduce: no output");
    MI355_CHECK_ARG(!y || weight, "allreduce: RMSNorm weight required with y");
    FusedParams p;
    p.ar = dev_view(a);
    p.x = (const f16*)x_f16; p.partials = partials; p.nsplit = nsplit; p.ld = ld; p.bias = (const f16*)bias;
    p.res_in = (const f16*)residual_in; p.res_out = (f16*)residual_out; p.weight = (const f16*)weight; p.y = (f16*)y;
    p.eps = eps; p.T = T; p.H = H; p.y_img_mblk = y_img_mblk; p.prepub = prepub ? 1 : 0;
    p.pf = (const uint32_t*)a->pf_ptr; p.pf_lines = (uint32_t)(a->pf_bytes >> 7); p.pf_sink = (uint32_t*)a->status + 32;
    a->pf_ptr = nullptr; a->pf_bytes = 0;                    // one launch only
    const int grid = T < kMaxBlocks ? T : kMaxBlocks;
    MI355_CHECK_ARG((size_t)cdiv(T, grid) * H * 2 <= a->slot_bytes, "allreduce: %d rows of %d per block exceed the %zu-byte slot", cdiv(T, grid), H, a->slot_bytes);
    hipStream_t st = (hipStream_t)stream;
    const bool two = T > kOneShotRows && a->world > 2;   // (N - 1) vs 2 (N - 1) / N reads per element: equal at N = 2
    if (a->ll && !a->full_fences && T <= kOneShotRows) {   // one row per block, granules.  The choice depends on context state only (every rank of a group holds the
        // same: distributed.CustomAllReduce agrees on it at creation) -- never on the prefetch pointer, which rides on the flag barrier and is simply unused here
#define LL_(V, B) hipLaunchKernelGGL((allreduce_fused_kernel<V, false, B, true>), dim3(grid), dim3(512), 0, st, p)
        if (act_dtype == MI355_ACT_BF16) { if (H / 8 <= 512) LL_(1, true); else LL_(2, true); }
        else                             { if (H / 8 <= 512) LL_(1, false); else LL_(2, false); }
#undef LL_
        MI355_CHECK_LAUNCH("allreduce_fused_kernel (LL)");
        return MI355_OK;
    }
#define L_(V, W, B) hipLaunchKernelGGL((allreduce_fused_kernel<V, W, B>), dim3(grid), dim3(512), 0, st, p)
    if (act_dtype == MI355_ACT_BF16) {
        if (H / 8 <= 512) { if (two) L_(1, true, true); else L_(1, false, true); }
        else              { if (two) L_(2, true, true); else L_(2, false, true); }
    } else {
        if (H / 8 <= 512) { if (two) L_(1, true, false); else L_(1, false, false); }
        else              { if (two) L_(2, true, false); else L_(2, false, false); }
    }
#undef L_
    MI355_CHECK_LAUNCH("allreduce_fused_kernel");
    return MI355_OK;
}

extern "C" int mi355_allreduce_sum(mi355_allreduce_t* a, const void* x_f16, void* out_f16, int32_t T, int32_t H,
                                   mi355_stream_t stream) {
    return mi355_allreduce_fused(a, x_f16, nullptr, 0, 0, nullptr, nullptr, out_f16, nullptr, 0.f, T, H, nullptr, stream);
}

extern "C" int mi355_allreduce_sum_dt(mi355_allreduce_t* a, const void* x, void* out, int32_t T, int32_t H, int32_t act_dtype,
                                      mi355_stream_t stream) {
    return mi355_allreduce_fused_dt(a, x, nullptr, 0, 0, nullptr, nullptr, out, nullptr, 0.f, T, H, nullptr, act_dtype, stream);
}

extern "C" int mi355_allgather_hidden(mi355_allreduce_t* a, const void* x_f16, void* out_f16, int32_t T, int32_t n,
                                      mi355_stream_t stream) {
    MI355_CHECK_ARG(a && a->ready, "allgather: context not opened (mi355_allreduce_open)");
    MI355_CHECK_ARG(x_f16 && out_f16 && x_f16 != out_f16, "allgather: null or aliased tensors");
    MI355_CHECK_ARG(T > 0 && n > 0 && n % 8 == 0, "allgather: T=%d n=%d (n %% 8 == 0)", T, n);
    const size_t bytes = (size_t)T * n * a->world * 2;
    MI355_CHECK_ARG(bytes <= a->max_bytes, "allgather: gathered tensor %zu bytes > registered %zu", bytes, a->max_bytes);
    GatherParams p;
    p.ar = dev_view(a);
    p.x = (const f16*)x_f16; p.out = (f16*)out_f16; p.T = T; p.n = n;
    const int grid = T < kMaxBlocks ? T : kMaxBlocks;
    MI355_CHECK_ARG((size_t)cdiv(T, grid) * n * a->world * 2 <= a->slot_bytes, "allgather: rows per block exceed the %zu-byte slot", a->slot_bytes);
    hipLaunchKernelGGL(allgather_hidden_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    MI355_CHECK_LAUNCH("allgather_hidden_kernel");
    return MI355_OK;
}

extern "C" int mi355_allreduce_argmax(mi355_allreduce_t* a, const float* logits, int32_t B, int32_t V_local, int32_t ld,
                                      int32_t vocab_offset, int32_t* ids, int32_t* positions, void* workspace,
                                      size_t workspace_bytes, mi355_stream_t stream) {
    MI355_CHECK_ARG(a && a->ready, "allreduce_argmax: context not opened");
    MI355_CHECK_ARG(logits && ids && workspace && B > 0 && V_local > 0 && ld >= V_local && ld % 4 == 0, "allreduce_argmax: bad args");
    MI355_CHECK_ARG((size_t)B * 8 <= kAuxBytes, "allreduce_argmax: batch %d too large for the record region", B);
    const int nparts = 64;
    if (workspace_bytes < (size_t)B * nparts * 8) { mi355_set_error("allreduce_argmax: workspace too small"); return MI355_ERR_WORKSPACE; }
    // stage 1 of the local argmax (candidates per row), then the cross-rank pick
    if (int e = mi355_argmax_candidates(logits, B, V_local, ld, workspace, workspace_bytes, stream)) return e;
    ArgmaxParams p;
    p.ar = dev_view(a);
    p.cand_v = (const float*)workspace; p.cand_i = (const int*)((const float*)workspace + (size_t)B * nparts);
    p.nparts = nparts; p.B = B; p.vocab_offset = vocab_offset; p.ids = ids; p.positions = positions;
    const int grid = B < kMaxBlocks ? B : kMaxBlocks;
    hipLaunchKernelGGL(allreduce_argmax_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, p);
    MI355_CHECK_LAUNCH("allreduce_argmax_kernel");
    return MI355_OK;
}
