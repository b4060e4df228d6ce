// Probe: practical HBM read ceiling on this MI355X -- a pure streaming read (16 B per lane, non-temporal, D loads in
// flight per wave), for several footprints and wave counts.  The sum is written once per block so the loads are kept.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int D>
__global__ __launch_bounds__(512) void stream_read(const u32x4* __restrict__ src, size_t n16, unsigned* out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    for (; i + (D - 1) * stride < n16; i += D * stride) {
        u32x4 v[D];
#pragma unroll
        for (int d = 0; d < D; ++d) v[d] = __builtin_nontemporal_load(src + i + d * stride);
#pragma unroll
        for (int d = 0; d < D; ++d) acc ^= v[d];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[blockIdx.x] = 1;
}

int main() {
    const size_t bytes = (size_t)2 << 30;
    u32x4* buf; unsigned* out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 1 << 20);
    hipMemset(buf, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t sizes[] = {(size_t)72 << 20, (size_t)256 << 20, (size_t)1 << 30, (size_t)2 << 30};
    const int grids[] = {256, 512, 1024, 2048};
    for (size_t sz : sizes)
        for (int g : grids) {
            stream_read<4><<<g, 512>>>(buf, sz / 16, out);
            hipEventRecord(e0);
            const int reps = sz > ((size_t)512 << 20) ? 5 : 20;
            for (int r = 0; r < reps; ++r) stream_read<4><<<g, 512>>>(buf, sz / 16, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("footprint %5zu MiB  grid %4d x 512 threads, 4 x 16 B in flight per lane: %8.1f us per pass  %7.1f GB/s\n", sz >> 20, g,
                   ms / reps * 1e3, sz / (ms / reps * 1e-3) / 1e9);
        }
    return 0;
}
