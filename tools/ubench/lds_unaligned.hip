// Microbenchmark: LDS read cost of 16 bytes per lane at lane-varying byte offsets (what a lane needs when quality bytes
// are staged in LDS by coalesced loads and then picked up at arbitrary alignment).
// Build: hipcc --offload-arch=gfx950 -O3 lds_unaligned.hip -o lds_unaligned ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
struct __attribute__((packed, aligned(1))) Q16 { uint32_t w[4]; };
struct __attribute__((packed, aligned(4))) Q16a4 { uint32_t w[4]; };
template <int MODE> __global__ __launch_bounds__(256) void k(uint32_t stride, uint32_t offs, int iters, uint32_t *out)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[16384];
    for (int i = threadIdx.x; i < 16384 / 4; i += 256) reinterpret_cast<uint32_t *>(lds)[i] = i * 2654435761u;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t acc = 0, a = wave * 4096u + lane * stride + offs;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t at = (a + j * 64u) & 16383u & ~(MODE == 0 ? 0u : (MODE == 1 ? 3u : 15u));
            const uint32_t at2 = at > 16384u - 16u ? 0u : at;
            if (MODE == 0) { Q16 v; __builtin_memcpy(&v, lds + at2, 16); acc += v.w[0] ^ v.w[1] ^ v.w[2] ^ v.w[3]; }
            else if (MODE == 1) { Q16a4 v; __builtin_memcpy(&v, lds + at2, 16); acc += v.w[0] ^ v.w[1] ^ v.w[2] ^ v.w[3]; }
            else { const uint4 v = *reinterpret_cast<const uint4 *>(lds + at2); acc += v.x ^ v.y ^ v.z ^ v.w; }
        }
        a += 1040u;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int MODE> void run(const char *name, uint32_t stride, uint32_t offs, uint32_t *o)
{
    const int iters = 500, blocks = 2048;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, stride, offs, iters, o);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double wl = (double)blocks * 4 * iters * 8;
    printf("%-64s %7.3f ms  %6.1f CU-cycles per wave-read of 16 B/lane\n", name, ms, ms * 1e-3 * 2.4e9 / (wl / 256));
}
int main()
{
    uint32_t *o; (void)hipMalloc(&o, 2048 * 256 * 4);
    run<2>("16-B aligned b128, stride 16", 16, 0, o);
    run<2>("16-B aligned b128, stride 32", 32, 0, o);
    run<1>("4-aligned 16 B (packed, aligned 4), stride 16 +4", 16, 4, o);
    run<1>("4-aligned 16 B, stride 20", 20, 0, o);
    run<0>("byte-aligned 16 B (packed), stride 16 +1", 16, 1, o);
    run<0>("byte-aligned 16 B, stride 17", 17, 0, o);
    run<0>("byte-aligned 16 B, stride 150", 150, 0, o);
    return 0;
}
