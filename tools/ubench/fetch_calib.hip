// Calibration of rocprofv3's FETCH_SIZE for k_pileup's access pattern: every byte of a buffer is read
// exactly once by unaligned 16-byte loads, a lane quad covering 64 consecutive bytes of one 150-byte
// "read", 16 quads of a wave on different reads.  Known bytes = n_reads * 150 (+ the 16-byte overhang
// per read).  Compare with 2 x FETCH_SIZE.   Build: hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

struct __attribute__((packed, aligned(1))) Q16 { uint32_t w[4]; };

// mode 0: the quad pattern above;  mode 1: plain coalesced 16 B per lane (the guide's calibrated case)
__global__ __launch_bounds__(256) void k_read(const uint8_t *buf, uint64_t n_reads, uint32_t read_len, int mode, uint32_t *out)
{
    uint32_t acc = 0;
    const uint64_t quad = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const uint32_t ql = threadIdx.x & 3u;
    if (mode == 0) {
        for (uint64_t r = quad; r < n_reads; r += ((uint64_t)gridDim.x * blockDim.x) >> 2) {
            const uint8_t *p = buf + r * read_len;
            for (uint32_t u = ql; u * 16u < read_len; u += 4u) {
                Q16 v;
                __builtin_memcpy(&v, p + 16u * u, 16);
                acc += v.w[0] ^ v.w[1] ^ v.w[2] ^ v.w[3];
            }
        }
    } else {
        const uint64_t n16 = n_reads * read_len / 16;
        for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
            const uint4 v = reinterpret_cast<const uint4 *>(buf)[i];
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char **argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const uint64_t n_reads = 9000000; const uint32_t read_len = 150;
    const size_t bytes = n_reads * read_len + 64;
    uint8_t *d; uint32_t *o;
    hipMalloc(&d, bytes); hipMalloc(&o, 4);
    hipMemset(d, 1, bytes);
    hipDeviceSynchronize();
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k_read, dim3(8192), dim3(256), 0, 0, d, n_reads, read_len, mode, o);
    hipDeviceSynchronize();
    printf("mode %d: bytes per launch %llu\n", mode, (unsigned long long)(n_reads * read_len));
    return 0;
}
