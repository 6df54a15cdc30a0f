// kernels.hip.h -- gfx950 (MI355X / CDNA4) kernels of the callable-loci engine.
//
// Data flow of one contig (all arrays resident in HBM, layouts in DESIGN.md section 3):
//
//   k_read_prep      per read: CIGAR walk -> end[r]; block partials of the per-read separable
//                    sums (contig_profiler.rs:74,79-82 via SURVEY 8a-7), max span, max end
//   k_window_bounds  per window of T reference positions: [lo,hi) range of reads that can touch it
//   k_pileup<T>      one workgroup per window: the three per-position counters of
//                    process_position (mod.rs:17-42) are built in LDS (never in HBM), classified
//                    (callable_profiler.rs:100-116) and written as one state byte per position
//   k_fin_windows /  run-boundary counts -> exclusive offsets (two-level scan); window / read
//   k_fin_summary    partials -> contig summary
//   k_rle_write      state bytes -> (start,end,state) intervals (callable_profiler.rs:122-155)
//
// Integer / byte work throughout: HBM-bound, no MFMA.  Wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace clk {

constexpr int kBlock = 256;          // threads per workgroup (4 waves)
constexpr int kPrepBlocks = 4096;    // grid of k_read_prep (grid-stride)
constexpr int kQualPad = 32;         // bytes of padding in front of / behind the quality array
constexpr uint32_t kLutSize = 65536; // low-MAPQ threshold table entries (raw depth 0..65535)

enum : uint32_t { kErrCigar = 1u, kErrRange = 2u };

// per-block output of k_read_prep
struct PrepPartial {
    unsigned long long sum_reflen;       // -> summed_coverage
    unsigned long long sum_mapq_reflen;  // -> summed_mapq
    uint32_t max_span;
    uint32_t max_end;
    uint32_t err;
    uint32_t pad;
};

// per-window output of k_pileup
struct WinPartial {
    unsigned long long cnt[6];       // state counts
    unsigned long long n_cov;        // positions with raw_depth > 0
    unsigned long long sum_qc;       // -> quality_bases
    unsigned long long sum_q;        // -> summed_baseq
    uint32_t n_inner;                // run boundaries strictly inside the window
    uint32_t max_raw;
};

// mirrors cl_contig_summary (include/callable_loci.h) + engine-private tail
struct DevSummary {
    unsigned long long state_counts[6];
    unsigned long long n_covered_bases;
    unsigned long long summed_coverage;
    unsigned long long summed_baseq;
    unsigned long long summed_mapq;
    unsigned long long quality_bases;
    unsigned long long extent;
    unsigned long long max_raw_depth;
    unsigned long long n_intervals;
    // private
    unsigned long long max_end;
    unsigned long long err;
};

struct Interval { uint32_t start, end, state; };

struct Opts {
    uint32_t min_depth;
    uint32_t max_depth;
    uint32_t min_mapq;
    uint32_t min_depth_for_low_mapq;
    uint32_t max_low_mapq;
    double   max_low_mapq_fraction;
    // byte-parallel "quality >= min_base_quality" constants (see qual_ge)
    uint32_t ge_add, ge_or, ge_and;
};

struct Reads {
    const int32_t  *pos;
    const uint8_t  *mapq;
    const uint32_t *cigar_off;
    const uint32_t *cigar;
    const unsigned long long *qual_off;
    const uint8_t  *qual;       // points kQualPad bytes into the allocation
    uint32_t n;
};

__device__ __forceinline__ bool op_match(uint32_t op) { return op == 0u || op == 7u || op == 8u; }
__device__ __forceinline__ bool op_del(uint32_t op) { return op == 2u || op == 3u; }
__device__ __forceinline__ bool op_ins(uint32_t op) { return op == 1u || op == 4u; }

// ---------------------------------------------------------------------------------------------
// wave / block reductions (wave64)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    for (int o = 32; o > 0; o >>= 1) { uint32_t t = __shfl_down(v, o, 64); v = t > v ? t : v; }
    return v;
}
__device__ __forceinline__ uint32_t wave_or_u32(uint32_t v)
{
    for (int o = 32; o > 0; o >>= 1) v |= __shfl_down(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------
// k_read_prep: one thread per read (grid-stride).
//   end[r] = pos + bam_cigar2rlen  (the pileup node span)
//   sum_reflen       = sum over reads of reflen                 == summed_coverage
//   sum_mapq_reflen  = sum over reads with mapq >= min_mapq of mapq*reflen == summed_mapq
// CIGAR shapes htslib's resolve_cigar2 asserts on / indexes out of bounds for are flagged.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_read_prep(Reads R, Opts o, uint32_t *__restrict__ end_out,
                                                       PrepPartial *__restrict__ part)
{
    __shared__ unsigned long long s_a[kBlock / 64], s_b[kBlock / 64];
    __shared__ uint32_t s_c[kBlock / 64], s_d[kBlock / 64], s_e[kBlock / 64];
    unsigned long long sum_len = 0, sum_mq = 0;
    uint32_t max_span = 0, max_end = 0, err = 0;
    for (uint32_t r = blockIdx.x * kBlock + threadIdx.x; r < R.n; r += gridDim.x * kBlock) {
        const uint32_t k0 = R.cigar_off[r], k1 = R.cigar_off[r + 1];
        unsigned long long reflen = 0;
        for (uint32_t k = k0; k < k1; ++k) {
            const uint32_t c = R.cigar[k], op = c & 15u, l = c >> 4;
            if (op_match(op) || op_del(op)) {
                reflen += l;
                if (l == 0) err |= kErrCigar;          // zero-length reference-consuming op
            }
        }
        // a read that reaches a column with a single non-match op is undefined in htslib
        if (reflen > 0 && k1 - k0 == 1 && !op_match(R.cigar[k0] & 15u)) err |= kErrCigar;
        const unsigned long long e = (unsigned long long)(uint32_t)R.pos[r] + reflen;
        if (e > 0xFFFF0000ull) { err |= kErrRange; }
        const uint32_t e32 = e > 0xFFFF0000ull ? (uint32_t)R.pos[r] : (uint32_t)e;
        end_out[r] = e32;
        const uint32_t span = e32 - (uint32_t)R.pos[r];
        sum_len += span;
        const uint32_t mq = R.mapq[r];
        if (mq >= o.min_mapq) sum_mq += (unsigned long long)mq * span;
        max_span = span > max_span ? span : max_span;
        max_end = e32 > max_end ? e32 : max_end;
    }
    sum_len = wave_sum_u64(sum_len);
    sum_mq = wave_sum_u64(sum_mq);
    max_span = wave_max_u32(max_span);
    max_end = wave_max_u32(max_end);
    err = wave_or_u32(err);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_a[wv] = sum_len; s_b[wv] = sum_mq; s_c[wv] = max_span; s_d[wv] = max_end; s_e[wv] = err;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        PrepPartial p;
        p.sum_reflen = 0; p.sum_mapq_reflen = 0; p.max_span = 0; p.max_end = 0; p.err = 0; p.pad = 0;
        for (int i = 0; i < kBlock / 64; ++i) {
            p.sum_reflen += s_a[i]; p.sum_mapq_reflen += s_b[i];
            p.max_span = s_c[i] > p.max_span ? s_c[i] : p.max_span;
            p.max_end = s_d[i] > p.max_end ? s_d[i] : p.max_end;
            p.err |= s_e[i];
        }
        part[blockIdx.x] = p;
    }
}

// ---------------------------------------------------------------------------------------------
// k_window_bounds: thread per window.  Reads are sorted by pos; a read can touch window
// [W, W+T) only if pos < W+T and pos + max_span > W.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lower_bound_pos(const int32_t *pos, uint32_t n, long long key)
{
    uint32_t lo = 0, hi = n;                       // first r with pos[r] >= key
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if ((long long)pos[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(kBlock) void k_window_bounds(Reads R, const PrepPartial *__restrict__ part,
                                                           uint32_t T, uint32_t n_win,
                                                           uint32_t *__restrict__ win_lo,
                                                           uint32_t *__restrict__ win_hi,
                                                           unsigned long long *__restrict__ win_q0,
                                                           uint32_t *__restrict__ err_flag)
{
    __shared__ uint32_t s_m[kBlock / 64];
    uint32_t m = 0;
    for (int i = threadIdx.x; i < kPrepBlocks; i += kBlock) { uint32_t v = part[i].max_span; m = v > m ? v : m; }
    m = wave_max_u32(m);
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    uint32_t max_span = 0;
    for (int i = 0; i < kBlock / 64; ++i) max_span = s_m[i] > max_span ? s_m[i] : max_span;
    const uint32_t w = blockIdx.x * kBlock + threadIdx.x;
    if (w >= n_win) return;
    const long long W = (long long)w * T;
    const uint32_t lo = lower_bound_pos(R.pos, R.n, W - (long long)max_span + 1);
    const uint32_t hi = lower_bound_pos(R.pos, R.n, W + (long long)T);
    win_lo[w] = lo;
    win_hi[w] = hi;
    win_q0[w] = R.qual_off[lo];                    // lo <= n: the offsets array has n+1 entries
    // k_pileup addresses the quality bytes of a window with 32-bit offsets
    if (hi > lo && R.qual_off[hi] - R.qual_off[lo] > 0xFFFF0000ull) atomicOr(err_flag, kErrRange);
}

// ---------------------------------------------------------------------------------------------
// byte-parallel threshold test.  x holds 4 quality bytes; returns 0x80 in each byte with
// quality >= min_base_quality (mod.rs:33).  Constants from make_ge_consts():
//   T == 0        : always                   ge_add = 0x80.., OR form
//   1 <= T <= 128 : hi(x) | (lo7(x) >= T)    ge_add = 128 - T, OR form
//   T >= 129      : hi(x) & (lo7(x) >= T-128) ge_add = 256 - T, AND form
// lo7 + ge_add never carries out of its byte (both <= 127 / 128+127 < 256).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t qual_ge(uint32_t x, const Opts &o)
{
    const uint32_t d = (x & 0x7f7f7f7fu) + o.ge_add;
    return ((d | (x & o.ge_or)) & (x | o.ge_and)) & 0x80808080u;
}

struct __attribute__((packed, aligned(1))) Q16 { uint32_t w[4]; };

struct PileupArgs {
    Reads R;
    Opts o;
    const uint32_t *end;          // per read
    const uint32_t *win_lo, *win_hi;
    const unsigned long long *win_q0;   // qual_off[win_lo[w]]
    const uint8_t  *ref;          // padded with 'N' up to n_win*T
    const uint32_t *lut;          // kLutSize entries: smallest low count that is "too many"
    uint8_t        *state;        // n_win*T bytes
    WinPartial     *winpart;
    uint32_t        extent;       // positions >= extent are not classified
    uint32_t        n_win;
    uint32_t        n_win8;       // ceil(n_win/8): XCD-contiguous window ranges
    // debug dumps (nullptr in production)
    uint32_t *dbg_raw, *dbg_qc, *dbg_low;
    uint32_t ablate;              // timing experiments only (env CL_ABLATE); 0 in production
};

// One 16-position unit of one M/=/X segment: quality bytes v (unit position i <-> byte i), valid
// positions pm (bit i).  Adds the pass bits to the counters and returns the sum of the passing
// qualities (contig_profiler.rs:68-70).  Branch-free: an invalid position simply adds 0.
//   8-bit mode : byte counter per position, 4 positions per LDS word; the word index is
//                XOR-swizzled (word 4u+jj lives at 4u+(jj^((u>>3)&3))) so that the lanes of a
//                wave, which all hold the same jj, spread over all 32 banks
//   32-bit mode: one word per position (windows with a column deeper than 255)
// ORF: min_base_quality <= 128, the test is hi(x) | (lo7(x) + (128-T) >= 128) (see qual_ge).
// 0x01 in every byte of xw that is a valid position (pm nibble jj) and passes the threshold
template <bool ORF>
__device__ __forceinline__ uint32_t pass_bytes(uint32_t xw, uint32_t pm, int jj, const Opts &o)
{
    const uint32_t vm = __umul24((pm >> (4 * jj)) & 15u, 0x204081u) & 0x01010101u;   // nibble -> 0x01 per valid byte
    if (ORF) return ((((xw & 0x7f7f7f7fu) + o.ge_add) | xw) >> 7) & vm;
    return (qual_ge(xw, o) >> 7) & vm;
}

template <bool ORF>
__device__ __forceinline__ uint32_t apply_unit8(const Q16 &v, uint32_t pm, uint32_t u,
                                                uint32_t *__restrict__ s_qc, const Opts &o)
{
    uint32_t sq = 0;
    // byte address of word 4u + (jj ^ rot) == (16u | 4rot) ^ 4jj   (rot = (u>>3)&3)
    const uint32_t a0 = (u << 4) | ((u >> 1) & 12u);
    uint8_t *qc8 = reinterpret_cast<uint8_t *>(s_qc);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const uint32_t xw = v.w[jj];
        const uint32_t inc = pass_bytes<ORF>(xw, pm, jj, o);
        atomicAdd(reinterpret_cast<uint32_t *>(qc8 + (a0 ^ (4u * jj))), inc);
        sq = __builtin_amdgcn_udot4(xw, inc, sq, false);      // += quality of every passing byte
    }
    return sq;
}

template <bool ORF>
__device__ __forceinline__ uint32_t apply_unit32(const Q16 &v, uint32_t pm, uint32_t u,
                                                 uint32_t *__restrict__ s_qc, const Opts &o)
{
    uint32_t sq = 0;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const uint32_t xw = v.w[jj];
        const uint32_t inc = pass_bytes<ORF>(xw, pm, jj, o);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if ((inc >> (8 * i)) & 1u) atomicAdd(&s_qc[(u << 4) + 4 * jj + i], 1u);
        sq = __builtin_amdgcn_udot4(xw, inc, sq, false);
    }
    return sq;
}

// per-lane view of one read while the window is processed
struct ReadCur {
    uint32_t x;        // reference position of the next CIGAR op
    uint32_t y;        // query position of the next CIGAR op
    uint32_t k, k1;    // next / end CIGAR index
    uint32_t qrel;     // offset of the read's first quality byte from the window's quality base
    uint32_t qlen;     // l_seq
    uint32_t end, mq;
    uint32_t c0;       // prefetched first CIGAR word
    bool     live;
};

// a segment of the LDS list as a lane quad sees it
struct SegView {
    uint32_t srel, trel, qoff, u1, ub;
};
__device__ __forceinline__ SegView seg_view(uint2 d, uint32_t ql)
{
    SegView s;
    s.srel = d.y & 0xFFFFu;
    s.trel = s.srel + (d.y >> 16) + 1u;
    s.qoff = d.x + (uint32_t)kQualPad - s.srel;      // + 16*u = byte offset of unit u from the padded base
    s.u1 = (s.trel - 1u) >> 4;
    s.ub = (s.srel >> 4) + ql;
    return s;
}
// loads of one trip: units ub, ub+4, ub+8 (a unit past the end is clamped onto the last one)
__device__ __forceinline__ void seg_load3(const SegView &s, uint32_t ub, const uint8_t *__restrict__ qbase,
                                          Q16 v[3], uint32_t uu[3])
{
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const uint32_t u = ub + 4u * i;
        uu[i] = u < s.u1 ? u : s.u1;
        __builtin_memcpy(&v[i], qbase + (s.qoff + (uu[i] << 4)), 16);
    }
}
template <bool ORF>
__device__ __forceinline__ uint32_t seg_apply3(const SegView &s, uint32_t ub, bool on, const Q16 v[3],
                                               const uint32_t uu[3], bool mode8, uint32_t *__restrict__ s_qc,
                                               const Opts &o)
{
    uint32_t sq = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const uint32_t ps = uu[i] << 4;
        const uint32_t vs = s.srel > ps ? s.srel - ps : 0u;
        const uint32_t ve = (s.trel - ps) < 16u ? (s.trel - ps) : 16u;
        const uint32_t m = ((1u << ve) - 1u) & ~((1u << vs) - 1u);
        const uint32_t pm = (on && ub + 4u * i <= s.u1) ? m : 0u;
        if (mode8) sq += apply_unit8<ORF>(v[i], pm, uu[i], s_qc, o);
        else sq += apply_unit32<ORF>(v[i], pm, uu[i], s_qc, o);
    }
    return sq;
}

#ifndef CL_MINWAVES
#define CL_MINWAVES 8
#endif
#ifndef CL_RPL
#define CL_RPL 1
#endif
template <int T, bool DEBUG, bool ORF>
__global__ __launch_bounds__(kBlock, CL_MINWAVES) void k_pileup(PileupArgs a)
{
    constexpr int PER = T / kBlock;                 // positions per thread in scan / classify
    static_assert(PER == 8 || PER == 16, "T must be 2048 or 4096");
    constexpr int kSegCap = T / 2;                  // uint2 segments that fit in s_b
    constexpr int kSegPerLane = kSegCap / kBlock;   // segments a lane may emit per round
    constexpr int kRegCap = kSegCap / (kBlock / 64);   // list entries per wave region
    constexpr int kRPL = CL_RPL;                    // reads per lane and pass
    constexpr uint32_t kLutLds = 256;
    // LDS: 2 x T words + T flag bytes.  s_a: raw depth (phases 1-2), then the qc counters
    // (phase 3).  s_b: low-mapq depth (phases 1-2), then the segment list (phase 3).
    __shared__ __attribute__((aligned(16))) uint32_t s_a[T];
    __shared__ __attribute__((aligned(16))) uint32_t s_b[T];
    __shared__ __attribute__((aligned(16))) uint8_t s_flag[T];   // bit0: raw>0, bit1: low-mapq rule fired
    __shared__ uint16_t s_lut[kLutLds];             // low-mapq thresholds for raw < 256 (0xFFFF = never)
    __shared__ uint32_t s_nseg[2][kBlock / 64];   // per round parity and wave: entries in the wave's region
    __shared__ uint32_t s_wraw[kBlock / 64], s_wlow[kBlock / 64], s_wmax[kBlock / 64];
    __shared__ uint8_t s_last[kBlock];
    __shared__ unsigned long long s_acc[10];        // cnt[6], n_cov, sum_qc, sum_q, n_inner

    // XCD-aware window order: blocks b, b+8, b+16.. share an XCD (round-robin dispatch), give
    // each XCD one contiguous range of windows so neighbouring windows share its L2.
    const uint32_t w = (blockIdx.x & 7u) * a.n_win8 + (blockIdx.x >> 3);
    if (w >= a.n_win) return;
    const uint32_t tid = threadIdx.x;
    const uint32_t W = w * (uint32_t)T;
    const uint32_t Wend = W + (uint32_t)T;
    const int lane = tid & 63, wv = tid >> 6;
    const uint32_t p0 = W + tid * PER;

    const uint32_t lo = a.win_lo[w], hi = a.win_hi[w];
    // all quality bytes of the reads [lo,hi) lie within 2^32 of qual_off[lo] (checked by
    // k_window_bounds), so they are addressed by 32-bit offsets from a uniform base.  The base
    // sits kQualPad bytes low so that the offset of a unit start never goes negative.
    const unsigned long long qwin = a.win_q0[w];
    const uint8_t *qbase = a.R.qual + qwin - kQualPad;

    // reference bytes of this thread's positions: needed last, requested first
    uint32_t refw[PER / 4];
#pragma unroll
    for (int i = 0; i < PER / 4; ++i) refw[i] = reinterpret_cast<const uint32_t *>(a.ref + p0)[i];

    // ---- phase 0: clear ----
    {
        const uint4 z = make_uint4(0, 0, 0, 0);
        uint4 *r4 = reinterpret_cast<uint4 *>(s_a), *l4 = reinterpret_cast<uint4 *>(s_b);
        for (int i = tid; i < T / 4; i += kBlock) { r4[i] = z; l4[i] = z; }
        if (tid < 10) s_acc[tid] = 0;
        if (tid < 2 * (kBlock / 64)) (&s_nseg[0][0])[tid] = 0;
        if (tid < kLutLds) { const uint32_t v = a.lut[tid]; s_lut[tid] = v > 0xFFFFu ? (uint16_t)0xFFFFu : (uint16_t)v; }
    }
    __syncthreads();

    // reads of the window are taken kRPL per lane; all their metadata loads are issued together
    const uint32_t n_pass = (hi - lo + kRPL * kBlock - 1) / (kRPL * kBlock);
    ReadCur rc[kRPL];
    auto load_reads = [&](uint32_t pass) {
#pragma unroll
        for (int i = 0; i < kRPL; ++i) {
            const uint32_t r = lo + (pass * kRPL + i) * kBlock + tid;
            rc[i].live = r < hi;
            rc[i].x = 0; rc[i].y = 0; rc[i].k = 0; rc[i].k1 = 0; rc[i].qrel = 0; rc[i].qlen = 0;
            rc[i].end = 0; rc[i].mq = 0; rc[i].c0 = 0;
            if (r < hi) {
                rc[i].x = (uint32_t)a.R.pos[r];
                rc[i].end = a.end[r];
                rc[i].mq = a.R.mapq[r];
                rc[i].k = a.R.cigar_off[r];
                rc[i].k1 = a.R.cigar_off[r + 1];
                const unsigned long long q0 = a.R.qual_off[r], q1 = a.R.qual_off[r + 1];
                rc[i].qrel = (uint32_t)(q0 - qwin);
                rc[i].qlen = (q1 - q0) > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)(q1 - q0);
            }
        }
    };

    // ---- phase 1: raw_depth and low_mapq_count as +1/-1 at the clipped read span ends ----
    // (both count every read whose [pos,end) covers the position, D/N included: mod.rs:22-28)
    for (uint32_t pass = 0; pass < ((a.ablate & 4u) ? 0u : n_pass); ++pass) {
        load_reads(pass);
#pragma unroll
        for (int i = 0; i < kRPL; ++i) {
            if (rc[i].live && rc[i].end > W) {
                const uint32_t cb = rc[i].x > W ? rc[i].x - W : 0u;
                const uint32_t ce = rc[i].end - W;
                atomicAdd(&s_a[cb], 1u);
                if (ce < (uint32_t)T) atomicAdd(&s_a[ce], 0xFFFFFFFFu);
                if (rc[i].mq <= a.o.max_low_mapq) {
                    atomicAdd(&s_b[cb], 1u);
                    if (ce < (uint32_t)T) atomicAdd(&s_b[ce], 0xFFFFFFFFu);
                }
            }
        }
    }
    // with a single pass the reads stay in registers for phase 3; their first CIGAR word is
    // requested now and arrives behind the scan
    auto prefetch_c0 = [&]() {
#pragma unroll
        for (int i = 0; i < kRPL; ++i) {
            rc[i].live = rc[i].live && rc[i].mq >= a.o.min_mapq && rc[i].end > W && rc[i].k < rc[i].k1;
            if (rc[i].live) rc[i].c0 = a.R.cigar[rc[i].k];
        }
    };
    if (n_pass == 1) prefetch_c0();
    __syncthreads();

    // ---- phase 2: prefix sums -> depths; per position keep only (raw>0, low-mapq rule) ----
    uint32_t maxraw;
    {
        uint32_t vr[PER], vl[PER];
        uint32_t sr = 0, sl = 0, mx = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            sr += s_a[tid * PER + i]; vr[i] = sr;
            sl += s_b[tid * PER + i]; vl[i] = sl;
        }
        uint32_t ir = sr, il = sl;
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t tr = __shfl_up(ir, o, 64), tl = __shfl_up(il, o, 64);
            if (lane >= o) { ir += tr; il += tl; }
        }
        if (lane == 63) { s_wraw[wv] = ir; s_wlow[wv] = il; }
        __syncthreads();                              // also: everyone has read s_a / s_b
        uint32_t offr = ir - sr, offl = il - sl;
        for (int i = 0; i < wv; ++i) { offr += s_wraw[i]; offl += s_wlow[i]; }
        uint32_t fl[PER / 4];
#pragma unroll
        for (int i = 0; i < PER / 4; ++i) fl[i] = 0;
        uint32_t ncov = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const uint32_t raw = vr[i] + offr, low = vl[i] + offl;
            mx = raw > mx ? raw : mx;
            // callable_profiler.rs:100-101
            bool is_low = false;
            if (raw >= a.o.min_depth_for_low_mapq && raw > 0) {
                if (raw < kLutLds) { const uint32_t th = s_lut[raw]; is_low = th != 0xFFFFu && low >= th; }
                else if (raw < kLutSize) is_low = low >= a.lut[raw];
                else is_low = ((double)low / (double)raw) > a.o.max_low_mapq_fraction;   // IEEE f64 divide
            }
            const uint32_t f = (raw > 0 ? 1u : 0u) | (is_low ? 2u : 0u);
            fl[i >> 2] |= f << (8 * (i & 3));
            if (p0 + i < a.extent) ncov += raw > 0 ? 1u : 0u;
            if (DEBUG) {
                if (a.dbg_raw) a.dbg_raw[p0 + i] = raw;
                if (a.dbg_low) a.dbg_low[p0 + i] = low;
            }
        }
#pragma unroll
        for (int i = 0; i < PER / 4; ++i) reinterpret_cast<uint32_t *>(s_flag)[tid * (PER / 4) + i] = fl[i];
        if (ncov) atomicAdd(&s_acc[6], (unsigned long long)ncov);
        mx = wave_max_u32(mx);
        if (lane == 0) s_wmax[wv] = mx;
        // s_a becomes the qc counter array
        const uint4 z = make_uint4(0, 0, 0, 0);
        uint4 *q4 = reinterpret_cast<uint4 *>(s_a);
        for (int i = tid; i < T / 4; i += kBlock) q4[i] = z;
        __syncthreads();
        maxraw = 0;
        for (int i = 0; i < kBlock / 64; ++i) maxraw = s_wmax[i] > maxraw ? s_wmax[i] : maxraw;
    }
    // qc_depth <= raw_depth at every position, so byte counters cannot overflow when the
    // window's largest column is <= 255; otherwise use one 32-bit counter per position.
    const bool mode8 = maxraw <= 255u;
    uint32_t *s_qc = s_a;
    uint2 *s_seg = reinterpret_cast<uint2 *>(s_b);   // {quality offset, srel | (len-1)<<16}

    // ---- phase 3: qc_depth -- M/=/X bases with base quality >= min (reads with mapq >= min) ----
    // Rounds of two steps.  A: each lane walks the CIGARs of its reads and appends the
    // window-clipped M/=/X segments (at most kSegPerLane per round) to the LDS list.  B: lane quads
    // take segments from the list, two at a time; a lane handles units of 16 reference positions
    // = one unaligned 16-byte load of quality bytes (six in flight), a byte-parallel threshold
    // test and packed LDS counter adds.
    unsigned long long sumq = 0;
    {
        uint32_t par = 0;
        const uint32_t ql = tid & 3u;
        for (uint32_t pass = 0; pass < ((a.ablate & 2u) ? 0u : n_pass); ++pass) {
            if (n_pass > 1) { load_reads(pass); prefetch_c0(); }
            bool first[kRPL];
#pragma unroll
            for (int i = 0; i < kRPL; ++i) first[i] = true;
            for (;;) {
                // -- A: emit segments --
                uint32_t nemit = 0;
#pragma unroll
                for (int i = 0; i < kRPL; ++i) {
                    ReadCur &c = rc[i];
                    while (c.live && nemit < (uint32_t)kSegPerLane) {
                        if (c.k >= c.k1 || c.x >= Wend) { c.live = false; break; }
                        const uint32_t cw = first[i] ? c.c0 : a.R.cigar[c.k];
                        first[i] = false;
                        const uint32_t op = cw & 15u, l = cw >> 4;
                        ++c.k;
                        if (op_match(op)) {
                            const uint32_t xe = c.x + l;
                            if (xe > W && c.y < c.qlen) {
                                const uint32_t s = c.x > W ? c.x : W;
                                uint32_t t = xe < Wend ? xe : Wend;
                                const uint32_t lq = (c.qlen - c.y) < l ? (c.qlen - c.y) : l;   // bases that have a quality byte
                                t = (c.x + lq) < t ? (c.x + lq) : t;
                                if (s < t) {
                                    const uint32_t slot = wv * kRegCap + atomicAdd(&s_nseg[par][wv], 1u);
                                    s_seg[slot] = make_uint2(c.qrel + c.y + (s - c.x), (s - W) | ((t - s - 1u) << 16));
                                    ++nemit;
                                }
                            }
                            c.x = xe; c.y += l;
                        } else if (op_del(op)) {
                            c.x += l;
                        } else if (op_ins(op)) {
                            c.y += l;
                        }
                    }
                    c.live = c.live && c.k < c.k1 && c.x < Wend;
                }
                bool any_live = false;
#pragma unroll
                for (int i = 0; i < kRPL; ++i) any_live = any_live || rc[i].live;
                const int more = __syncthreads_or(any_live ? 1 : 0);
                // -- B: consume segments --
                // Each wave's region of the list is in read (= position) order.  The 16 quads of a
                // wave take entries that are far apart (4 regions x 4 strided sub-ranges), so that one
                // wave instruction never adds to the same counter word from several lanes.
                if (tid < kBlock / 64) s_nseg[par ^ 1u][tid] = 0;
                uint32_t sq32 = 0;
                {
                    const uint32_t qw = (tid >> 2) & 15u, region = qw & 3u, sub = qw >> 2;
                    const uint32_t nr = (a.ablate & 1u) ? 0u : s_nseg[par][region];
                    const uint32_t Q = (nr + 3u) >> 2;
                    const uint2 *reg = s_seg + region * kRegCap;
                    for (uint32_t rr = wv; rr < Q; rr += kBlock / 64) {
                        const uint32_t idx = sub * Q + rr;
                        if (idx < nr) {
                            const SegView sa = seg_view(reg[idx], ql);
                            for (uint32_t u = sa.ub; u <= sa.u1; u += 12u) {
                                Q16 va[3];
                                uint32_t ua[3];
                                seg_load3(sa, u, qbase, va, ua);
                                sq32 += seg_apply3<ORF>(sa, u, true, va, ua, mode8, s_qc, a.o);
                            }
                        }
                    }
                }
                sumq += sq32;
                __syncthreads();
                par ^= 1u;
                if (!more) break;
            }
        }
    }

    // ---- phase 4: classify (callable_profiler.rs:104-116), count, write state bytes ----
    {
        uint32_t st[PER];
        unsigned long long cntp = 0;                // six 8-bit fields (PER <= 16)
        unsigned long long sqc = 0;
        uint32_t qcw[PER / 4], flw[PER / 4];
#pragma unroll
        for (int i = 0; i < PER / 4; ++i) {
            const uint32_t wi = tid * (PER / 4) + i;          // word 4u+jj with u = wi>>2, jj = wi&3
            flw[i] = reinterpret_cast<const uint32_t *>(s_flag)[wi];
            qcw[i] = mode8 ? s_qc[(wi & ~3u) | ((wi & 3u) ^ ((wi >> 5) & 3u))] : 0u;
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const uint32_t p = p0 + i;
            const uint32_t f = (flw[i >> 2] >> (8 * (i & 3))) & 0xFFu;
            const uint32_t qc = mode8 ? ((qcw[i >> 2] >> (8 * (i & 3))) & 0xFFu) : s_qc[tid * PER + i];
            if (DEBUG) { if (a.dbg_qc) a.dbg_qc[p] = qc; }
            uint32_t s = 0xFFu;
            if (p < a.extent) {
                const uint32_t rb = (refw[i >> 2] >> (8 * (i & 3))) & 0xFFu;
                if (rb == 'N' || rb == 'n') s = 0u;                                  // REF_N
                else if (!(f & 1u)) s = 2u;                                          // NO_COVERAGE
                else if (f & 2u) s = 5u;                                             // POOR_MAPPING_QUALITY
                else if (qc < a.o.min_depth) s = 3u;                                 // LOW_COVERAGE
                else if (a.o.max_depth > 0 && qc > a.o.max_depth) s = 4u;            // EXCESSIVE_COVERAGE
                else s = 1u;                                                         // CALLABLE
                cntp += 1ull << (8u * s);
                sqc += qc;
            }
            st[i] = s;
        }
        // run boundaries strictly inside the window: position p (> W) whose state differs from p-1
        s_last[tid] = (uint8_t)st[PER - 1];
        __syncthreads();
        uint32_t nb = 0;
        uint32_t prev = tid > 0 ? s_last[tid - 1] : st[0];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            if (p0 + i < a.extent && st[i] != prev) nb += 1;
            prev = st[i];
        }
        if (PER == 8) {
            uint2 v;
            v.x = st[0] | (st[1] << 8) | (st[2] << 16) | (st[3] << 24);
            v.y = st[4] | (st[5] << 8) | (st[6] << 16) | (st[7] << 24);
            *reinterpret_cast<uint2 *>(a.state + p0) = v;
        } else {
            uint4 v;
            v.x = st[0] | (st[1] << 8) | (st[2] << 16) | (st[3] << 24);
            v.y = st[4] | (st[5] << 8) | (st[6] << 16) | (st[7] << 24);
            v.z = st[8 % PER] | (st[9 % PER] << 8) | (st[10 % PER] << 16) | (st[11 % PER] << 24);
            v.w = st[12 % PER] | (st[13 % PER] << 8) | (st[14 % PER] << 16) | (st[15 % PER] << 24);
            *reinterpret_cast<uint4 *>(a.state + p0) = v;
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const uint32_t v = (uint32_t)(cntp >> (8 * c)) & 0xFFu;
            if (v) atomicAdd(&s_acc[c], (unsigned long long)v);
        }
        if (sqc) atomicAdd(&s_acc[7], sqc);
        if (sumq) atomicAdd(&s_acc[8], sumq);
        if (nb) atomicAdd(&s_acc[9], (unsigned long long)nb);
    }
    __syncthreads();
    if (tid == 0) {
        WinPartial wp;
        for (int c = 0; c < 6; ++c) wp.cnt[c] = s_acc[c];
        wp.n_cov = s_acc[6]; wp.sum_qc = s_acc[7]; wp.sum_q = s_acc[8];
        wp.n_inner = (uint32_t)s_acc[9];
        wp.max_raw = maxraw;
        a.winpart[w] = wp;
    }
}

// ---------------------------------------------------------------------------------------------
// k_fin_windows / k_fin_summary: two-level exclusive scan of the run starts per window (inner
// boundaries + the seam with the previous window) and reduction of the window / read partials
// to the contig summary.
// ---------------------------------------------------------------------------------------------
constexpr int kFinBlock = 1024;

struct FinPartial {
    unsigned long long acc[9];       // cnt[6], n_cov, sum_qc, sum_q
    uint32_t n_runs;
    uint32_t max_raw;
};

__global__ __launch_bounds__(kFinBlock) void k_fin_windows(const WinPartial *__restrict__ winpart,
                                                            const uint8_t *__restrict__ state, uint32_t T,
                                                            uint32_t n_win, uint32_t extent,
                                                            uint32_t *__restrict__ win_off,
                                                            FinPartial *__restrict__ fin)
{
    __shared__ uint32_t s_w[kFinBlock / 64], s_m[kFinBlock / 64];
    __shared__ unsigned long long s_red[9][kFinBlock / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t w = blockIdx.x * kFinBlock + tid;
    unsigned long long acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t c = 0, maxraw = 0;
    if (w < n_win) {
        const WinPartial wp = winpart[w];
        c = wp.n_inner;
        const uint32_t p = w * T;
        if (p < extent) c += (w == 0) ? 1u : (state[p] != state[p - 1] ? 1u : 0u);
        for (int i = 0; i < 6; ++i) acc[i] = wp.cnt[i];
        acc[6] = wp.n_cov; acc[7] = wp.sum_qc; acc[8] = wp.sum_q;
        maxraw = wp.max_raw;
    }
    uint32_t inc = c;
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) s_w[wv] = inc;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const unsigned long long v = wave_sum_u64(acc[i]);
        if (lane == 0) s_red[i][wv] = v;
    }
    maxraw = wave_max_u32(maxraw);
    if (lane == 0) s_m[wv] = maxraw;
    __syncthreads();
    uint32_t off = inc - c;
    for (int i = 0; i < wv; ++i) off += s_w[i];
    if (w < n_win) win_off[w] = off;                 // relative to this block's first window
    if (tid == 0) {
        FinPartial fp;
        fp.n_runs = 0; fp.max_raw = 0;
        for (int j = 0; j < kFinBlock / 64; ++j) { fp.n_runs += s_w[j]; fp.max_raw = s_m[j] > fp.max_raw ? s_m[j] : fp.max_raw; }
        for (int i = 0; i < 9; ++i) { unsigned long long v = 0; for (int j = 0; j < kFinBlock / 64; ++j) v += s_red[i][j]; fp.acc[i] = v; }
        fin[blockIdx.x] = fp;
    }
}

__global__ __launch_bounds__(kBlock) void k_fin_summary(const FinPartial *__restrict__ fin, uint32_t n_fin,
                                                         const PrepPartial *__restrict__ prep, uint32_t n_prep,
                                                         uint32_t extent, const uint32_t *__restrict__ err_flag,
                                                         uint32_t *__restrict__ blk_off,
                                                         DevSummary *__restrict__ out)
{
    __shared__ unsigned long long s_red[11][kBlock / 64];
    __shared__ uint32_t s_u[3][kBlock / 64];
    __shared__ uint32_t s_w[kBlock / 64];
    __shared__ uint32_t s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    unsigned long long acc[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t maxraw = 0, maxend = 0, err = 0;
    for (uint32_t base = 0; base < n_fin; base += kBlock) {
        const uint32_t b = base + tid;
        uint32_t c = 0;
        if (b < n_fin) {
            const FinPartial fp = fin[b];
            c = fp.n_runs;
            for (int i = 0; i < 9; ++i) acc[i] += fp.acc[i];
            maxraw = fp.max_raw > maxraw ? fp.max_raw : maxraw;
        }
        uint32_t inc = c;
        for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        if (lane == 63) s_w[wv] = inc;
        __syncthreads();
        uint32_t off = s_carry + inc - c;
        for (int i = 0; i < wv; ++i) off += s_w[i];
        if (b < n_fin) blk_off[b] = off;
        __syncthreads();
        if (tid == kBlock - 1) s_carry = off + c;
        __syncthreads();
    }
    for (uint32_t i = tid; i < n_prep; i += kBlock) {
        acc[9] += prep[i].sum_reflen; acc[10] += prep[i].sum_mapq_reflen;
        maxend = prep[i].max_end > maxend ? prep[i].max_end : maxend; err |= prep[i].err;
    }
#pragma unroll
    for (int i = 0; i < 11; ++i) {
        const unsigned long long v = wave_sum_u64(acc[i]);
        if (lane == 0) s_red[i][wv] = v;
    }
    maxraw = wave_max_u32(maxraw); maxend = wave_max_u32(maxend); err = wave_or_u32(err);
    if (lane == 0) { s_u[0][wv] = maxraw; s_u[1][wv] = maxend; s_u[2][wv] = err; }
    __syncthreads();
    if (tid == 0) {
        unsigned long long tot[11];
        for (int i = 0; i < 11; ++i) { tot[i] = 0; for (int j = 0; j < kBlock / 64; ++j) tot[i] += s_red[i][j]; }
        uint32_t mr = 0, me = 0, er = 0;
        for (int j = 0; j < kBlock / 64; ++j) { mr = s_u[0][j] > mr ? s_u[0][j] : mr; me = s_u[1][j] > me ? s_u[1][j] : me; er |= s_u[2][j]; }
        for (int i = 0; i < 6; ++i) out->state_counts[i] = tot[i];
        out->n_covered_bases = tot[6];
        out->quality_bases = tot[7];
        out->summed_baseq = tot[8];
        out->summed_coverage = tot[9];
        out->summed_mapq = tot[10];
        out->extent = extent;
        out->max_raw_depth = mr;
        out->n_intervals = s_carry;
        out->max_end = me;
        out->err = er | *err_flag;
    }
}

// ---------------------------------------------------------------------------------------------
// k_rle_write: one workgroup per window; recomputes run starts from the state bytes and writes
// the intervals.  The thread that finds the start of run i also closes run i-1.
// ---------------------------------------------------------------------------------------------
template <int T>
__global__ __launch_bounds__(kBlock) void k_rle_write(const uint8_t *__restrict__ state,
                                                       const uint32_t *__restrict__ win_off,
                                                       const uint32_t *__restrict__ blk_off,
                                                       uint32_t n_win, uint32_t extent,
                                                       Interval *__restrict__ iv, uint32_t iv_cap)
{
    constexpr int PER = T / kBlock;
    __shared__ uint32_t s_w[kBlock / 64];
    const uint32_t w = blockIdx.x;
    if (w >= n_win) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t p0 = w * (uint32_t)T + tid * PER;
    uint8_t st[PER];
    if (PER >= 16) {
#pragma unroll
        for (int i = 0; i < PER; i += 16) {
            const uint4 v = *reinterpret_cast<const uint4 *>(state + p0 + i);
            const uint32_t ww[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 16; ++j) st[i + j] = (uint8_t)(ww[j >> 2] >> (8 * (j & 3)));
        }
    } else {
        const uint2 v = *reinterpret_cast<const uint2 *>(state + p0);
        const uint32_t ww[2] = {v.x, v.y};
#pragma unroll
        for (int j = 0; j < 8; ++j) st[j] = (uint8_t)(ww[j >> 2] >> (8 * (j & 3)));
    }
    uint32_t prev = p0 > 0 ? state[p0 - 1] : 0x100u;   // position 0 always starts a run
    uint32_t flags = 0, c = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        if (p0 + i < extent && st[i] != prev) { flags |= 1u << i; ++c; }
        prev = st[i];
    }
    uint32_t inc = c;
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    uint32_t idx = blk_off[w / kFinBlock] + win_off[w] + inc - c;
    for (uint32_t i = 0; i < wv; ++i) idx += s_w[i];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        if (flags & (1u << i)) {
            if (idx < iv_cap) { iv[idx].start = p0 + i; iv[idx].state = st[i]; }
            if (idx > 0 && idx - 1 < iv_cap) iv[idx - 1].end = p0 + i;
            ++idx;
        }
    }
    // the thread that owns the last classified position closes the last run
    if (extent > 0 && p0 <= extent - 1 && extent - 1 < p0 + PER) {
        if (idx > 0 && idx - 1 < iv_cap) iv[idx - 1].end = extent;
    }
}

// ---------------------------------------------------------------------------------------------
// config 5: site-list pileup (src/haplogroup/caller.rs:62-152).  Thread per read; for every
// M/=/X base whose 1-based position is a listed site, hist[site][4-bit code] += 1.
// site_of: dense map position(0-based) -> site index or 0xFFFFFFFF, length map_len.
// ---------------------------------------------------------------------------------------------
struct SiteReads {
    const int32_t  *pos;
    const uint8_t  *mapq;
    const uint32_t *cigar_off;
    const uint32_t *cigar;
    const unsigned long long *seq_off;
    const uint8_t  *seq4;
    uint32_t n;
};

__global__ __launch_bounds__(kBlock) void k_site_pileup(SiteReads R, uint32_t min_quality, uint32_t contig_len,
                                                         unsigned long long ref_len,
                                                         const uint32_t *__restrict__ sorted_pos0,
                                                         const uint32_t *__restrict__ sorted_idx,
                                                         uint32_t n_sites, uint32_t *__restrict__ hist)
{
    for (uint32_t r = blockIdx.x * kBlock + threadIdx.x; r < R.n; r += gridDim.x * kBlock) {
        if ((uint32_t)R.pos[r] >= contig_len) continue;          // fetch("chr:1-len"), caller.rs:33-36
        if ((uint32_t)R.mapq[r] < min_quality) continue;         // caller.rs:80
        unsigned long long x = (uint32_t)R.pos[r];
        unsigned long long y = 0;
        const unsigned long long s0 = R.seq_off[r], slen = R.seq_off[r + 1] - s0;
        for (uint32_t k = R.cigar_off[r]; k < R.cigar_off[r + 1]; ++k) {
            const uint32_t c = R.cigar[k], op = c & 15u, l = c >> 4;
            if (op_match(op)) {
                // sites with 0-based position in [x, x+l): binary search the first one
                uint32_t lo = 0, hi = n_sites;
                while (lo < hi) { uint32_t m = lo + ((hi - lo) >> 1); if (sorted_pos0[m] < x) lo = m + 1; else hi = m; }
                for (; lo < n_sites && sorted_pos0[lo] < x + l; ++lo) {
                    const unsigned long long p = sorted_pos0[lo];
                    const unsigned long long qi = y + (p - x);
                    if (qi < slen && p < ref_len) {               // caller.rs:105,110-113
                        const unsigned long long bi = s0 + qi;
                        const uint32_t byte = R.seq4[bi >> 1];
                        const uint32_t code = (bi & 1ull) ? (byte & 15u) : (byte >> 4);
                        atomicAdd(&hist[(unsigned long long)sorted_idx[lo] * 16ull + code], 1u);
                    }
                }
                x += l; y += l;
            } else if (op_del(op)) {
                x += l;
            } else if (op_ins(op)) {
                y += l;
            }
        }
    }
}

} // namespace clk
