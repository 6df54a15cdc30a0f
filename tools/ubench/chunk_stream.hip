// Microbenchmark: HBM read rate when every workgroup streams its data as `nchunk` interleaved chunk streams of `chunk`
// bytes each taken from far-apart places (what a long-read window's qualities look like: ~60 reads, 2 KB of each),
// against one contiguous stream of the same size.  16-byte per-lane loads, 4 in flight.
// Build: hipcc --offload-arch=gfx950 -O3 chunk_stream.hip -o chunk_stream ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(256) void k(const uint8_t *buf, size_t bytes, uint32_t chunk, uint32_t per_wg, uint32_t *out)
{
    // workgroup b reads per_wg bytes: chunk c of it lives at ((b * 7919 + c * 104729) % n_slots) * chunk
    const size_t n_slots = bytes / chunk;
    const uint32_t nchunk = per_wg / chunk;
    uint32_t acc = 0;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    // a wave-load covers 1 KB; the four waves take consecutive KB of a chunk
    for (uint32_t c = 0; c < nchunk; ++c) {
        const size_t slot = chunk == per_wg ? (size_t)blockIdx.x : (((size_t)blockIdx.x * 7919u + (size_t)c * 104729u) % n_slots);
        const uint8_t *p = buf + slot * chunk;
        for (uint32_t off = wave * 1024u; off < chunk; off += 4096u) {
            const uint4 v = *reinterpret_cast<const uint4 *>(p + off + lane * 16u);
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main()
{
    const size_t bytes = (size_t)3 << 30;
    uint8_t *d; uint32_t *o;
    (void)hipMalloc(&d, bytes); (void)hipMalloc(&o, 32768 * 256 * 4);
    (void)hipMemset(d, 1, bytes);
    const uint32_t per_wg = 128u << 10;                 // a window's worth of qualities
    const uint32_t blocks = (uint32_t)(bytes / per_wg);
    for (uint32_t chunk : {128u << 10, 32u << 10, 8u << 10, 4u << 10, 2u << 10, 1u << 10}) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, bytes, chunk, per_wg, o);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        printf("chunks of %6u B: %7.3f ms  %5.2f TB/s\n", chunk, ms, (double)blocks * per_wg / (ms * 1e-3) / 1e12);
    }
    return 0;
}
