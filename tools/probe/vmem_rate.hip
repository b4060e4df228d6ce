// Probe (next round's first GPU call): what does ONE vector-memory wave-instruction cost a CU, by kind, and what is a CU's fill
// rate from L2?  Every wave issues N loads of the given kind in batches of 8 (8 in flight per wave) and the launch is timed with
// hipEvents; cycles per instruction per CU = time x clock x CUs / total instructions (one block per CU, W waves per block).
//   kind 0  dense 1 KB (16 B per lane) from a 64 KB window per block: L2 hits after the first pass   -> fill rate L2 -> CU
//   kind 1  the same instruction with every lane out of range of its buffer descriptor (no traffic)  -> pure issue cost
//   kind 2  4 lanes of 64 in range (the one-row activation fragment of the full-K kernels)
//   kind 3  4-byte loads, 16 distinct dwords per wave (the per-tile meta load)
//   kind 4  dense 1 KB, all blocks read the SAME 64 KB window (broadcast out of L2)
// usage: vmem_rate            (prints a table; ~2 s)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(1024) void vmem_rate(const char* __restrict__ base, int n_batches, unsigned* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
    const char* win = base + (KIND == 4 ? 0 : (size_t)blockIdx.x * 65536);
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)win, 0, 65536, 0x00020000u);
    unsigned voff;
    if (KIND == 1) voff = 0x80000000u;
    else if (KIND == 2) voff = (lane & 15) == 0 ? (unsigned)(lane >> 4) * 16u : 0x80000000u;
    else if (KIND == 3) voff = (unsigned)(lane & 15) * 4u;
    else voff = (unsigned)lane * 16u;
    u32x4 acc = {0, 0, 0, 0};
    unsigned soff = (unsigned)wave * 1024u;
    for (int b = 0; b < n_batches; ++b) {
        u32x4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned so = (soff + (unsigned)i * (unsigned)W * 1024u) & 0xFFFFu & ~1023u;
            if (KIND == 3) { v[i] = (u32x4){(unsigned)__builtin_amdgcn_raw_buffer_load_b32(r, voff, so, 0), 0u, 0u, 0u}; }
            else v[i] = __builtin_amdgcn_raw_buffer_load_b128(r, voff, so, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc ^= v[i];
        soff += 8u * (unsigned)W * 1024u;
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[blockIdx.x] = 1;
}

template <int KIND>
static void run(const char* buf, unsigned* out, int grid, int waves, const char* name) {
    const int n_batches = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    vmem_rate<KIND><<<grid, waves * 64>>>(buf, n_batches, out);
    hipEventRecord(e0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) vmem_rate<KIND><<<grid, waves * 64>>>(buf, n_batches, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms / reps * 1e3, instr_per_cu = (double)waves * n_batches * 8;
    int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    const double cyc = us * 1e-6 * clk_khz * 1e3 / instr_per_cu;
    const double gbs = (KIND == 0 || KIND == 4) ? instr_per_cu * 1024 / (us * 1e-6) / 1e9 : 0.0;
    printf("%-34s grid %3d x %2d waves: %8.1f us  %6.1f cycles per wave-instruction per CU (at %d MHz)%s", name, grid, waves, us, cyc, clk_khz / 1000,
           gbs > 0 ? "" : "\n");
    if (gbs > 0) printf("  %6.1f GB/s per CU, %5.2f TB/s chip\n", gbs, gbs * grid / 1e3);
}

int main() {
    char* buf; unsigned* out;
    hipMalloc(&buf, (size_t)256 * 65536); hipMalloc(&out, 1 << 16);
    hipMemset(buf, 1, (size_t)256 * 65536);
    for (int grid : {1, 256})
        for (int waves : {1, 4, 8, 16}) {
            run<0>(buf, out, grid, waves, "dense 1 KB, L2-resident window");
            run<1>(buf, out, grid, waves, "all lanes out of range");
            run<2>(buf, out, grid, waves, "4 of 64 lanes in range");
            run<3>(buf, out, grid, waves, "4-byte loads, 16 distinct dwords");
            run<4>(buf, out, grid, waves, "dense 1 KB, one window for all");
        }
    return 0;
}
