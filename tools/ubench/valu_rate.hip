// Microbenchmark: issue cost of the integer VALU / LDS instructions k_pileup is made of.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstdlib>

#define REP 64
template <int OP> __global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t a0, uint32_t b0, int iters)
{
    __shared__ unsigned long long lds[2048];
    uint32_t x[8];
    for (int i = 0; i < 8; ++i) x[i] = a0 + threadIdx.x * (i + 1);
    uint32_t b = b0 | 1u;
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = 0;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) x[i] = (x[i] & 0x7f7f7f7fu) + b;                       // and + add (2 instr)
                else if (OP == 1) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
                else if (OP == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
                else if (OP == 3) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x80" : "+v"(x[i]) : "v"(b));
                else if (OP == 4) asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(x[i]) : "v"(b));
                else if (OP == 5) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[i]) : "v"(b));
                else if (OP == 6) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(b));
                else if (OP == 7) asm volatile("v_lshrrev_b32 %0, 7, %0" : "+v"(x[i]));
                else if (OP == 8) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
                else if (OP == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(b));
                else if (OP == 14) asm volatile("v_lerp_u8 %0, %0, %1, %1" : "+v"(x[i]) : "v"(b));
                else if (OP == 15) asm volatile("v_sad_u8 %0, %0, %1, %0" : "+v"(x[i]) : "v"(b));
                else if (OP == 16) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(b));
                else if (OP == 17) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x[i]) : "v"(b));
                else if (OP == 18) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(b));
                else if (OP == 19) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(x[i]));
                else if (OP == 20) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
                else if (OP == 21) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(x[i]) : "v"(b));
                else if (OP == 11) atomicAdd(&lds[(threadIdx.x * 2 + i) & 2047], (unsigned long long)x[i]);          // ds_add_u64 conflict-free-ish
                else if (OP == 12) atomicAdd(reinterpret_cast<uint32_t *>(lds) + ((threadIdx.x + 64 * i) & 4095), x[i]); // ds_add_u32 conflict-free
                else if (OP == 13) x[i] += (uint32_t)lds[(threadIdx.x + i * 7) & 2047];                               // ds_read_b64
            }
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + (uint32_t)lds[threadIdx.x];
}

template <int OP> void run(const char *name, int instr_per_rep, uint32_t *d)
{
    const int iters = 200, blocks = 256 * 8;           // 8 blocks of 4 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 3u, 5u, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 3u, 5u, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * 4 * iters * REP * instr_per_rep;      // wave-instructions
    const double per_simd_per_s = winstr / 1024 / (ms * 1e-3);
    printf("%-14s %8.3f ms  %6.2f G wave-instr/s/SIMD  -> %5.2f cycles/instr at 2.4 GHz, %5.2f at 2.1\n", name, ms,
           per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s, 2.1e9 / per_simd_per_s);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    uint32_t *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    const int which = argc > 1 ? atoi(argv[1]) : -1;
    switch (which) {
    case 0: run<0>("and+add", 2, d); break;
    case 1: run<1>("v_and", 1, d); break;
    case 2: run<2>("v_add", 1, d); break;
    case 3: run<3>("v_bitop3", 1, d); break;
    case 4: run<4>("v_dot4_u8", 1, d); break;
    case 5: run<5>("v_mul_u24", 1, d); break;
    case 6: run<6>("v_perm", 1, d); break;
    case 7: run<7>("v_lshr", 1, d); break;
    case 8: run<8>("v_mul_lo_u32", 1, d); break;
    case 9: run<9>("v_cndmask", 1, d); break;
    case 14: run<14>("v_lerp_u8", 1, d); break;
    case 15: run<15>("v_sad_u8", 1, d); break;
    case 16: run<16>("v_and_or", 1, d); break;
    case 17: run<17>("v_lshl_add", 1, d); break;
    case 18: run<18>("v_add3", 1, d); break;
    case 19: run<19>("v_bfe", 1, d); break;
    case 20: run<20>("v_min_u32", 1, d); break;
    case 21: run<21>("v_alignbit", 1, d); break;
    case 11: run<11>("ds_add_u64", 1, d); break;
    case 12: run<12>("ds_add_u32", 1, d); break;
    case 13: run<13>("ds_read_b64", 1, d); break;
    default:
        if (which == -1) {      // every case, one process each would be cleaner; the kernels are independent
            run<1>("v_and", 1, d); run<2>("v_add", 1, d); run<3>("v_bitop3", 1, d); run<4>("v_dot4_u8", 1, d); run<5>("v_mul_u24", 1, d);
            run<6>("v_perm", 1, d); run<7>("v_lshr", 1, d); run<9>("v_cndmask", 1, d); run<14>("v_lerp_u8", 1, d); run<15>("v_sad_u8", 1, d);
            run<16>("v_and_or", 1, d); run<17>("v_lshl_add", 1, d); run<18>("v_add3", 1, d); run<19>("v_bfe", 1, d); run<20>("v_min_u32", 1, d);
            run<21>("v_alignbit", 1, d); run<11>("ds_add_u64", 1, d); run<12>("ds_add_u32", 1, d); run<13>("ds_read_b64", 1, d);
        } else printf("usage: valu_rate [case]\n");
    }
    hipDeviceSynchronize();
    return 0;
}
