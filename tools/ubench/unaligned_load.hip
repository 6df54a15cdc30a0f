// Microbenchmark: chip-wide rate of per-lane 16-byte loads whose addresses advance by `stride` bytes per lane
// (16 = aligned and contiguous; 17..31 = the unaligned, overlapping pattern k_pileup's quality loads have), with
// `inflight` loads issued per lane before their data is used.
// Build: hipcc --offload-arch=gfx950 -O3 unaligned_load.hip -o unaligned_load ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
struct __attribute__((packed, aligned(1))) Q16 { uint32_t w[4]; };
template <int INF> __global__ __launch_bounds__(256) void k(const uint8_t *buf, size_t bytes, uint32_t stride, uint32_t *out)
{
    // every workgroup streams its own contiguous slice, a wave 64 * stride bytes per load instruction
    const size_t per_wg = bytes / gridDim.x;
    const uint8_t *base = buf + (size_t)blockIdx.x * per_wg;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const size_t step = (size_t)64 * stride * 4;                 // the four waves interleave
    uint32_t acc = 0;
    for (size_t off = (size_t)wave * 64 * stride; off + (size_t)64 * stride * INF * 4 + 64 < per_wg; off += step * INF) {
        Q16 v[INF];
#pragma unroll
        for (int j = 0; j < INF; ++j) __builtin_memcpy(&v[j], base + off + j * step + (size_t)lane * stride, 16);
#pragma unroll
        for (int j = 0; j < INF; ++j) acc += v[j].w[0] ^ v[j].w[1] ^ v[j].w[2] ^ v[j].w[3];
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main()
{
    const size_t bytes = (size_t)3 << 30;
    uint8_t *d; uint32_t *o;
    hipMalloc(&d, bytes + 4096); hipMalloc(&o, 2048 * 256 * 4);
    hipMemset(d, 1, bytes + 4096);
    for (uint32_t stride : {16u, 17u, 20u, 24u, 32u}) {
        for (int inf : {2, 4}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (inf == 2) hipLaunchKernelGGL(k<2>, dim3(2048), dim3(256), 0, 0, d, bytes, stride, o);
                else hipLaunchKernelGGL(k<4>, dim3(2048), dim3(256), 0, 0, d, bytes, stride, o);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double lane_loads = (double)bytes / stride;
            printf("stride %2u  in flight %d:  %7.3f ms   %6.2f TB/s of distinct bytes   %6.1f G lane-loads/s   %5.2f lane-loads/ns/CU\n", stride, inf, ms,
                   bytes / (ms * 1e-3) / 1e12, lane_loads / (ms * 1e-3) / 1e9, lane_loads / (ms * 1e-3) / 1e9 / 256);
        }
    }
    return 0;
}
