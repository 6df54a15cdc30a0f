// Microbenchmark: rate of per-lane 16-byte global loads served from L2 (a 2 MB footprint read over and over), by
// address pattern -- what the vector memory pipe of a CU sustains when HBM is not the limit.
// Build: hipcc --offload-arch=gfx950 -O3 ta_rate.hip -o ta_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
struct __attribute__((packed, aligned(1))) Q16 { uint32_t w[4]; };
struct __attribute__((packed, aligned(1))) Q8 { uint32_t w[2]; };
template <int BYTES> __global__ __launch_bounds__(256) void k(const uint8_t *buf, uint32_t foot, uint32_t stride, uint32_t offs, int iters, uint32_t *out)
{
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
    uint32_t acc = 0;
    uint32_t a = (wave * 64u * stride * 7u) % foot;
    for (int it = 0; it < iters; ++it) {
        uint32_t x[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t at = ((a + j * 64u * stride) % foot) + lane * stride + offs;
            if (BYTES == 16) { Q16 v; __builtin_memcpy(&v, buf + at, 16); x[j][0] = v.w[0]; x[j][1] = v.w[1]; x[j][2] = v.w[2]; x[j][3] = v.w[3]; }
            else { Q8 v; __builtin_memcpy(&v, buf + at, 8); x[j][0] = v.w[0]; x[j][1] = v.w[1]; x[j][2] = 0; x[j][3] = 0; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += x[j][0] ^ x[j][1] ^ x[j][2] ^ x[j][3];
        a = (a + 4u * 64u * stride) % foot;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
// groups of G consecutive lanes read G * 16 contiguous bytes; the groups of a wave sit at scattered places (multiples of
// `gran` bytes plus `offs`): what lane quads that take consecutive units of one read's qualities look like
template <int G> __global__ __launch_bounds__(256) void kg(const uint8_t *buf, uint32_t foot, uint32_t gran, uint32_t offs, int iters, uint32_t *out)
{
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
    const uint32_t grp = lane / G, in = lane % G;
    uint32_t acc = 0, h = wave * 2654435761u + grp * 40503u;
    for (int it = 0; it < iters; ++it) {
        uint32_t x[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h = h * 1664525u + 1013904223u;
            const uint32_t at = ((h >> 8) % (foot / gran)) * gran + offs + in * 16u;
            Q16 v; __builtin_memcpy(&v, buf + at, 16); x[j][0] = v.w[0]; x[j][1] = v.w[1]; x[j][2] = v.w[2]; x[j][3] = v.w[3];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += x[j][0] ^ x[j][1] ^ x[j][2] ^ x[j][3];
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
// lanes at irregular increasing addresses: lane i at sum of steps drawn from [lo, hi] (in units of `mul` bytes), masked
// to `amask`, plus offs: what consecutive pieces of a read's qualities look like
__global__ __launch_bounds__(256) void ki(const uint8_t *buf, uint32_t foot, uint32_t lo, uint32_t hi, uint32_t mul, uint32_t amask, uint32_t offs, int iters, uint32_t *out)
{
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
    // per-lane prefix of pseudo-random steps (fixed per lane: the pattern, not the data, is what is measured)
    uint32_t pre = 0;
    for (uint32_t i = 0; i < lane; ++i) pre += (lo + ((i * 2654435761u >> 13) % (hi - lo + 1))) * mul;
    uint32_t acc = 0, a = (wave * 4099u * 64u) % foot;
    for (int it = 0; it < iters; ++it) {
        uint32_t x[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t at = (((a + j * 1536u) % foot) + pre) & amask;
            Q16 v; __builtin_memcpy(&v, buf + at + offs, 16); x[j][0] = v.w[0]; x[j][1] = v.w[1]; x[j][2] = v.w[2]; x[j][3] = v.w[3];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += x[j][0] ^ x[j][1] ^ x[j][2] ^ x[j][3];
        a = (a + 4u * 1536u) % foot;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
void runi(const char *name, const uint8_t *d, uint32_t foot, uint32_t lo, uint32_t hi, uint32_t mul, uint32_t amask, uint32_t offs, uint32_t *o)
{
    const int iters = 400, blocks = 2048;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(ki, dim3(blocks), dim3(256), 0, 0, d, foot, lo, hi, mul, amask, offs, iters, o);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double wl = (double)blocks * 4 * iters * 4;
    printf("%-58s %7.3f ms  %6.1f cycles per wave-load per CU\n", name, ms, ms * 1e-3 * 2.4e9 / (wl / 256));
}
template <int G> void rung(const char *name, const uint8_t *d, uint32_t foot, uint32_t gran, uint32_t offs, uint32_t *o)
{
    const int iters = 400, blocks = 2048;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kg<G>, dim3(blocks), dim3(256), 0, 0, d, foot, gran, offs, iters, o);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double wl = (double)blocks * 4 * iters * 4;
    printf("%-58s %7.3f ms  %6.1f cycles per wave-load per CU  %5.2f lane-loads/ns/CU\n", name, ms, ms * 1e-3 * 2.4e9 / (wl / 256), wl * 64 / (ms * 1e-3) / 1e9 / 256);
}
int main()
{
    const uint32_t foot = 2u << 20;
    uint8_t *d; uint32_t *o;
    (void)hipMalloc(&d, foot + (1u << 17)); (void)hipMalloc(&o, 2048 * 256 * 4);
    (void)hipMemset(d, 1, foot + (1u << 17));
    struct Case { const char *name; int bytes; uint32_t stride, offs; } cases[] = {
        {"16 B aligned, stride 16", 16, 16, 0}, {"16 B at +1 (byte-unaligned), stride 16", 16, 16, 1}, {"16 B at +4 (dword-aligned), stride 16", 16, 16, 4},
        {"16 B at +8, stride 16", 16, 16, 8}, {"16 B, stride 17", 16, 17, 0}, {"16 B, stride 20", 16, 20, 0}, {"16 B, stride 8 (overlapping)", 16, 8, 0},
        {"16 B, stride 0 (one address)", 16, 0, 0}, {"16 B, stride 64 (a line per 2 lanes)", 16, 64, 0}, {"8 B aligned, stride 8", 8, 8, 0}, {"8 B at +1, stride 8", 8, 8, 1},
        {"8 B, stride 9", 8, 9, 0}};
    const int iters = 400, blocks = 2048;
    for (const Case &c : cases) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            if (c.bytes == 16) hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(256), 0, 0, d, foot, c.stride, c.offs, iters, o);
            else hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(256), 0, 0, d, foot, c.stride, c.offs, iters, o);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        }
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double wl = (double)blocks * 4 * iters * 4;          // wave-level load instructions
        printf("%-42s %7.3f ms  %6.1f cycles per wave-load per CU (2.4 GHz)  %5.2f lane-loads/ns/CU  %6.2f TB/s requested\n", c.name, ms,
               ms * 1e-3 * 2.4e9 / (wl / 256), wl * 64 / (ms * 1e-3) / 1e9 / 256, wl * 64 * c.bytes / (ms * 1e-3) / 1e12);
    }
    runi("irregular steps of 12/16/20 B (dword-aligned)", d, foot, 3, 5, 4, ~0u, 0, o);
    runi("irregular steps of 8..24 B in dwords (dword-aligned)", d, foot, 2, 6, 4, ~0u, 0, o);
    runi("irregular steps of 13..19 B, rounded down to dwords", d, foot, 13, 19, 1, ~3u, 0, o);
    runi("irregular steps of 13..19 B (byte-unaligned)", d, foot, 13, 19, 1, ~0u, 0, o);
    runi("irregular steps of 12/16/20 B, +1", d, foot, 3, 5, 4, ~0u, 1, o);
    runi("irregular steps of 4..60 B in dwords", d, foot, 1, 15, 4, ~0u, 0, o);
    runi("steps of 16 B exactly (control)", d, foot, 4, 4, 4, ~0u, 0, o);
    rung<4>("quads of 64 B at scattered 64-B lines (16-aligned)", d, foot, 64, 0, o);
    rung<4>("quads of 64 B at scattered places, +4 (dword-aligned)", d, foot, 64, 4, o);
    rung<4>("quads of 64 B at scattered places, +1 (byte-unaligned)", d, foot, 64, 1, o);
    rung<4>("quads of 64 B at scattered 16-B places", d, foot, 16, 0, o);
    rung<2>("pairs of 32 B at scattered 16-B places", d, foot, 16, 0, o);
    rung<2>("pairs of 32 B at scattered places, +4", d, foot, 16, 4, o);
    rung<2>("pairs of 32 B at scattered places, +1", d, foot, 16, 1, o);
    rung<1>("single 16 B at scattered 16-B places", d, foot, 16, 0, o);
    rung<1>("single 16 B at scattered places, +4", d, foot, 16, 4, o);
    rung<1>("single 16 B at scattered places, +1", d, foot, 16, 1, o);
    rung<16>("16 lanes x 256 B at scattered 256-B places", d, foot, 256, 0, o);
    rung<16>("16 lanes x 256 B at scattered places, +4", d, foot, 256, 4, o);
    rung<16>("16 lanes x 256 B at scattered places, +1", d, foot, 256, 1, o);
    return 0;
}
