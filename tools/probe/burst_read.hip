// Probe: how long does ONE launch take to read a cold 68 MB (gate_up W4) / 34 MB (down) / 8 MB (qkv) window of HBM, as a
// function of who reads what?  Every launch reads a fresh window of a 3 GB buffer (nothing left in L2 / Infinity Cache).
//   pattern 0  chip-wide sweep: at step s the whole grid reads one contiguous run (wave-load = 1 KB, grid x waves KB per step)
//   pattern 1  block-private contiguous region, the block's waves interleaved KB by KB
//   pattern 2  block-private region, every wave its own contiguous run of it (the full-K kernel's blocked K slices)
// D wave-loads (1 KB each) in flight per wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int D>
__global__ __launch_bounds__(1024) void burst_read(const char* __restrict__ base, size_t bytes, int pattern, unsigned* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = blockDim.x >> 6, G = gridDim.x;
    const size_t region = bytes / G / 1024 * 1024;           // per block
    const size_t per_wave = region / W / 1024;                // KB per wave
    size_t start, step;
    if (pattern == 0)      { start = ((size_t)blockIdx.x * W + wave) * 1024; step = (size_t)G * W * 1024; }
    else if (pattern == 1) { start = blockIdx.x * region + (size_t)wave * 1024; step = (size_t)W * 1024; }
    else                   { start = blockIdx.x * region + wave * per_wave * 1024; step = 1024; }
    const char* p = base + start + lane * 16;
    u32x4 acc = {0, 0, 0, 0};
    size_t i = 0;
    for (; i + D <= per_wave; i += D) {
        u32x4 v[D];
#pragma unroll
        for (int d = 0; d < D; ++d) v[d] = __builtin_nontemporal_load((const u32x4*)(p + (i + d) * step));
#pragma unroll
        for (int d = 0; d < D; ++d) acc ^= v[d];
    }
    for (; i < per_wave; ++i) acc ^= __builtin_nontemporal_load((const u32x4*)(p + i * step));
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[blockIdx.x] = 1;
}

int main() {
    const size_t total = (size_t)3 << 30;
    char* buf; unsigned* out;
    if (hipMalloc(&buf, total) != hipSuccess) return 1;
    hipMalloc(&out, 1 << 20);
    hipMemset(buf, 1, total);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t sizes[] = {(size_t)68 << 20, (size_t)34 << 20, (size_t)8 << 20};
    struct Cfg { int grid, waves; } cfgs[] = {{256, 16}, {256, 8}, {512, 8}, {474, 7}, {1024, 4}, {224, 15}, {144, 14}};
    for (size_t sz : sizes)
        for (Cfg c : cfgs)
            for (int pat = 0; pat < 3; ++pat)
                for (int D : {2, 10}) {
                    const int reps = 24;
                    size_t off = 0;
                    auto launch = [&]() {
                        if (off + sz > total) off = 0;
                        if (D == 2) burst_read<2><<<c.grid, c.waves * 64>>>(buf + off, sz, pat, out);
                        else        burst_read<10><<<c.grid, c.waves * 64>>>(buf + off, sz, pat, out);
                        off += ((sz + ((size_t)2 << 20) - 1) >> 21) << 21;
                    };
                    for (int r = 0; r < 4; ++r) launch();
                    hipEventRecord(e0);
                    for (int r = 0; r < reps; ++r) launch();
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    printf("%3zu MB  grid %4d x %2d waves  pattern %d  %2d KB in flight per wave: %7.2f us per launch  %7.1f GB/s\n", sz >> 20,
                           c.grid, c.waves, pat, D, ms / reps * 1e3, sz / (ms / reps * 1e-3) / 1e9);
                }
    return 0;
}
