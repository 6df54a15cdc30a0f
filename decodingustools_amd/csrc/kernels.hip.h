// kernels.hip.h -- gfx950 (MI355X / CDNA4) kernels of the callable-loci engine.
//
// Data flow of one contig (all arrays resident in HBM, layouts in DESIGN.md section 3):
//
//   (host, at upload: per window of T reference positions the [lo,hi) range of candidates that can touch it -- an
//                    index of the resident layout, like the offsets; WinMeta below)
//   (host, at upload, in the one walk over every CIGAR that validates a tile: every read's end, and for reads of
//                    more than kLongOps operations a (reference, query) checkpoint before every 64th operation, where
//                    the host's later walks enter such a read)
//   (host, at upload: short-read contigs become 16-byte RECORDS, a head per read and a piece per further M/=/X run
//                    (ReadRec); long-read contigs a table of match pieces per window (run table) -- k_pileup decodes
//                    no CIGAR in either form, and no CIGAR is uploaded)
//   k_pileup<T>      one workgroup per window: the three per-position counters of
//                    process_position (mod.rs:17-42) are built in LDS (never in HBM), classified
//                    (callable_profiler.rs:100-116) and reduced to the window's run list (the
//                    positions inside the window where the state changes) and its totals
//   k_fin_windows    run counts per window (inner starts + the seam with the previous window) ->
//                    exclusive offsets inside blocks of kFinBlock windows
//   k_rle_write      run lists -> (start,end,state) intervals (callable_profiler.rs:122-155), one wave
//                    per window; its last workgroup reduces the partials to the contig summary
//
// Integer / byte work throughout: HBM-bound, no MFMA.  Wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace clk {

constexpr int kBlock = 256;          // threads per workgroup (4 waves)
constexpr uint32_t kLongOps = 64;    // reads with more CIGAR ops get a checkpoint (reference, query position) before
                                     // every 64th op (op numbering of the contig's CIGAR array): host side only
constexpr int kQualPad = 32;         // bytes of padding in front of / behind the quality array
constexpr uint32_t kWideSpan = 16384; // reads spanning more reference than this are "wide": looked up per window
                                      // in their own list instead of widening every window's candidate range
constexpr uint32_t kLutSize = 65536; // low-MAPQ threshold table entries (raw depth 0..65535)

enum : uint32_t { kErrCigar = 1u, kErrRange = 2u, kNeedDeep = 4u, kNeedWide8 = 8u };

// per-window output of k_pileup
struct WinPartial {
    unsigned long long cnt[6];       // state counts
    unsigned long long n_cov;        // positions with raw_depth > 0
    unsigned long long sum_qc;       // -> quality_bases
    unsigned long long sum_q;        // -> summed_baseq
    unsigned long long sum_reflen;   // reads that START in this window: sum of reference spans (-> summed_coverage) ...
    unsigned long long sum_mapq_reflen;   // ... and of mapq * span over those with mapq >= min (-> summed_mapq)
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
    // byte-parallel "quality >= min_base_quality" constants (see pass_bytes)
    uint32_t ge_k, ge_c;
    // the same for "qc_depth >= min_depth" (md_all: min_depth > 255, every byte-sized count is below)
    // and "qc_depth >= max_depth + 1" (xd_on: max_depth in 1..254), used by the byte-parallel final phase
    uint32_t md_add, md_or, md_and, md_all;
    uint32_t xd_add, xd_or, xd_and, xd_on;
};

struct Reads {
    const int32_t  *pos;        // run-table form: the windows' candidates (+-1 span scatter, owner sums)
    const uint8_t  *mapq;
    const uint8_t  *qual;       // points kQualPad bytes into the allocation
    uint32_t n;
};

// The short-read form of k_pileup reads RECORDS, 16 bytes each, one aligned load; the host builds them at upload
// (callable_loci.hip: gen_read_recs) in read order, the records of a read side by side:
//   head record  {pos, span, qual_lo, mapq | 0x100 | seglen << 16}: the read as the pileup holds it, [pos, pos + span)
//                (span = bam_cigar2rlen: D and N included) -- the +-1 scatter and, in the window that holds pos, the
//                separable sums.  When the read's first M/=/X run starts at pos (the usual case) the head carries it too:
//                seglen bases whose quality bytes start at qual_lo; else seglen = 0.
//   piece record {pos of the run, 0, qual_lo, mapq | seglen << 16}: one further M/=/X run (or the next 65 535 bases of
//                a longer one), clipped to the bases that have a quality byte.
// qual_lo = the low 32 bits of the run's quality offset: a window's candidates lie within 2^32 bytes of its q0.
// A read without a reference span has no record at all.
struct __attribute__((aligned(16))) ReadRec {
    int32_t  pos;
    uint32_t span;
    uint32_t qual_lo;
    uint32_t meta;
};

// The pass-bit form (k_pileup_rows) reads HEADS, 8 bytes each: {pos, span | low << 31} -- the read as the pileup holds
// it, [pos, pos + span) (span = bam_cigar2rlen: D and N included; mod.rs:22-28), and whether its mapq is at or below
// max_low_mapq (mod.rs:26-28).  Nothing else of a read is needed there: its M/=/X bases are in the rows, and its shares of
// summed_coverage and summed_mapq (contig_profiler.rs:74, 79-82: per-read separable) are added up by
// cl_push_reads' walk on the host.  A span of more than kHeadSpanMax positions is cut into several heads (the +-1
// scatter of [a, b) and [b, c) is that of [a, c)); a read without a reference span has none.
constexpr uint32_t kHeadSpanMax = 0x7FFFFFFFu;

// 16 bytes at any byte address (compiles to one unaligned dwordx4 load)
struct __attribute__((packed, aligned(1))) Q16 { uint32_t w[4]; };

__device__ __forceinline__ bool op_match(uint32_t op) { return op == 0u || op == 7u || op == 8u; }
__device__ __forceinline__ bool op_del(uint32_t op) { return op == 2u || op == 3u; }
__device__ __forceinline__ bool op_ins(uint32_t op) { return op == 1u || op == 4u; }

// ---------------------------------------------------------------------------------------------
// wave / block reductions and scans (wave64)
// ---------------------------------------------------------------------------------------------
// Inclusive prefix sum over the 64 lanes with DPP row shifts / row broadcasts (VALU only; __shfl_up
// compiles to ds_bpermute_b32, an LDS-pipe instruction with far higher latency).
//   row_shr:1,2,4,8 (0x111..0x118, zero fill) scan each row of 16 lanes,
//   row_bcast:15 (0x142, rows 1 and 3) adds the previous row's total,
//   row_bcast:31 (0x143, rows 2 and 3) adds the total of lanes 0..31.
__device__ __forceinline__ uint32_t dpp_incl_scan_u32(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}
// inclusive prefix sum within each row of 16 lanes (the first four steps of the above)
__device__ __forceinline__ uint32_t dpp_row_incl_scan_u32(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    return v;
}
// the maximum over the wave, in every lane (same DPP pattern; the zero fill is the identity of an unsigned max)
__device__ __forceinline__ uint32_t dpp_wave_max_u32(uint32_t v)
{
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); v = t > v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); v = t > v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); v = t > v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); v = t > v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); v = t > v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); v = t > v ? t : v;
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// the sum over the wave, in every lane (lane 63 of the scan, read through an SGPR)
__device__ __forceinline__ uint32_t dpp_wave_sum_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)dpp_incl_scan_u32(v), 63);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) { return dpp_wave_max_u32(v); }
__device__ __forceinline__ uint32_t wave_or_u32(uint32_t v)
{
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// 64-bit sum over the wave, in every lane: three exact 32-bit DPP sums over 24/24/16-bit limbs
// (64 lanes x 2^24 < 2^32), instead of twelve ds_bpermute round trips
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
    const unsigned long long a = dpp_wave_sum_u32((uint32_t)v & 0xFFFFFFu);
    const unsigned long long b = dpp_wave_sum_u32((uint32_t)(v >> 24) & 0xFFFFFFu);
    const unsigned long long c = dpp_wave_sum_u32((uint32_t)(v >> 48));
    return a + (b << 24) + (c << 48);
}

// What k_pileup needs to start on a window: one 32-byte record, one scalar load.  The candidate reads of a window --
// reads [lo, hi) with pos < W + T and pos + span_n > W (span_n: the longest ordinary span), and the wide reads (span >
// kWideSpan) that start before them -- depend on the resident reads and the extent only: the host builds the records at
// upload (callable_loci.hip: host_window_bounds), flags windows whose candidates' quality bytes do not fit 32-bit
// offsets (kErrRange) and windows with more than 32 767 candidates (kNeedDeep: the 32-bit counter form).
struct __attribute__((aligned(32))) WinMeta {
    uint32_t lo, hi;                   // ordinary candidates: reads [lo, hi)
    uint32_t wlo, wn;                  // wide candidates: wide_idx[wlo .. wlo+wn)
    unsigned long long q0;             // qual_off of the window's first candidate read
    uint32_t rlo, rn;                  // run-table form: the window's entries are runtab[rlo .. rlo + rn)
};

// ---------------------------------------------------------------------------------------------
// byte-parallel ">= threshold" on four bytes at once, given the three constants of make_ge_consts()
// (callable_loci.hip) for a threshold T: 0x80 in each byte >= T.
//   T == 0        : always                    add = 0x80.., OR form
//   1 <= T <= 128 : hi(x) | (lo7(x) >= T)     add = 128 - T, OR form
//   T >= 129      : hi(x) & (lo7(x) >= T-128) add = 256 - T, AND form
// lo7 + add never carries out of its byte (both <= 127 / 128+127 < 256).  Used by the final phase for
// "qc_depth >= min_depth" / "> max_depth"; the quality threshold itself is pass_bytes() below.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t swar_ge7(uint32_t q, uint32_t add, uint32_t orm, uint32_t andm)
{
    const uint32_t d = (q & 0x7f7f7f7fu) + add;
    return ((d | (q & orm)) & (q | andm)) & 0x80808080u;
}


struct PileupArgs {
    Reads R;
    Opts o;
    const ReadRec *rec;           // the records of the short-read form (it reads these and nothing else per read)
    const uint2   *heads;         // pass-bit form: {pos, span | low << 31} per read with a reference span
    const uint32_t *end;          // per read, from the host (run-table form)
    const WinMeta *win;
    const uint32_t *wide_idx;           // read indices of the wide reads, ascending
    const uint8_t  *ref;          // padded with 'N' up to n_win*T (byte forms)
    const uint32_t *refn;         // pass-bit form: bit p = the reference base at p is 'N' / 'n' (or beyond the reference)
    const uint32_t *lut;          // kLutSize entries: smallest low count that is "too many"
    const uint2    *runtab;       // run-table form (LONG = 2): per window, the M/=/X pieces of its reads (host, at upload)
    const uint4    *rows;         // pass-bit form (k_pileup_rows): per window, groups of 4 rows x 64 blocks (host, at upload)
    uint8_t        *state;        // n_win*T bytes; written by the DEBUG instantiation only (test dumps)
    uint16_t       *runs;         // per window T entries: the run starts strictly inside the window, rel. position | state << 12
    uint8_t        *first_state, *last_state;   // per window: state of its first / last position (run seams)
    WinPartial     *winpart;
    uint32_t        extent;       // positions >= extent are not classified
    uint32_t        n_win;
    uint32_t        n_win8;       // ceil(n_win/8): XCD-contiguous window ranges
    // debug dumps (nullptr in production)
    uint32_t *dbg_raw, *dbg_qc, *dbg_low;
    uint32_t ablate;              // timing experiments only (CL_TUNING builds: env CL_ABLATE); ignored otherwise
    uint32_t upl;                 // quality units per lane and trip in the consume loop: 2 for reads of up to ~128 bases, else 3
    uint8_t  *win_wide;           // per window: 1 = a position deeper than 255 was seen here, use 16-bit fields (sticky
                                  // for the resident contig; set by k_pileup itself, see mode8 below)
    uint32_t *err_flag;           // kNeedWide8 is raised here
};

// ---------------------------------------------------------------------------------------------
// k_pileup: one workgroup per window of T reference positions.
//
// Pass over the window's candidates (LONG = 0: the records of its reads, ReadRec; the wide reads' that start before
// the ordinary range first), 256 at a time, one lane per candidate, waves never synchronising:
//   * +1/-1 at the clipped span ends into raw / low-mapq difference arrays (mod.rs:22-28: every
//     read covering a position counts, D/N included)
//   * the lane writes the window-clipped M/=/X segment of its record (mapq >= min_mapq) into its wave's private
//     LDS list (in lane = position order); no CIGAR is decoded -- the host's walk at upload made the records
//   * lane quads consume the list: a lane handles units of 16 reference positions = one unaligned
//     16-byte load of quality bytes, a byte-parallel "quality >= min" test (mod.rs:30-37) and
//     adds into packed 8-bit (two sets) or 16-bit LDS counters (qc_depth); the sum of the passing
//     qualities feeds summed_baseq (contig_profiler.rs:68-70)
// then one barrier and a final phase per position: prefix sums -> raw_depth / low_mapq_count,
// the low-MAPQ rule and the state (callable_profiler.rs:100-116), the window's totals and its run
// list (the positions inside the window where the state changes).
// Neither the per-position counters nor the per-position states ever exist in HBM.
//
// Candidates are dealt to waves round-robin (candidate = base + 4*lane + wave): a wave's list holds every fourth
// read, and consecutive candidates alternate between the two 8-bit counter sets.
//
// LONG = 2 (the run-table form; what a contig with 8 or more CIGAR operations per read gets -- indel-rich ONT-like
// reads and HiFi-like long match runs alike; the operation-parallel form that decoded CIGARs on the device, LONG = 1 of
// rounds 1-3, lost to it on both and is gone): the host's walk over the CIGARs at upload leaves, per window, a flat
// table of the M/=/X pieces of its reads --
// 8 bytes each: {quality offset, window-relative start | end - 1 | counter set}, a piece never longer than two
// 16-position units, reads below min_mapq already dropped -- and the kernel streams its window's entries coalesced,
// one entry per lane, two entries and four quality loads in flight per lane.  The +-1 span scatter and the owner sums
// take pos / end / mapq of the window's candidates.
//
// DEEP = false: 8/16-bit counters and 16-bit differences; valid while the window has <= 32767 candidates
// (otherwise host_window_bounds raises kNeedDeep and the contig runs with DEEP = true: one
// 32-bit counter per position).
// ---------------------------------------------------------------------------------------------
template <bool ORF>
__device__ __forceinline__ uint32_t pass_bytes(uint32_t xw, uint32_t vm, const Opts &o)
{
    // 0x01 in every byte of xw that is a valid position (vm) and passes the threshold (mod.rs:33).  v_lerp_u8 is a
    // per-byte (x + k + (c & 1)) >> 1 with a 9-bit sum: with k = 256 - min_base_quality its bit 7 is the carry, i.e.
    // x >= min_base_quality, for every threshold 1..255 (0: k = 255 and the rounding bit make it always set) -- one
    // instruction where the masked add needs three (measured: 4.5 against 3 x 2.8 cycles per wave instruction)
    return (__builtin_amdgcn_lerp(xw, o.ge_k, o.ge_c) >> 7) & vm;
}

// 8-bit counters, two sets (reads alternate between the sets, a window handled this way is
// touched by <= 510 reads, so no byte exceeds 255): positions 8e..8e+7 are one 8-byte entry
// e = 2u + h of set `set`, stored at 2u + (h ^ ((u>>3)&1)); one ds_add_u64 covers 8 positions.
template <bool ORF>
__device__ __forceinline__ uint32_t apply_unit8(const Q16 &v, const uint4 vm, uint32_t u, uint32_t set_off,
                                                unsigned long long *__restrict__ s_qc, const Opts &o)
{
    const uint32_t i0 = pass_bytes<ORF>(v.w[0], vm.x, o), i1 = pass_bytes<ORF>(v.w[1], vm.y, o);
    const uint32_t i2 = pass_bytes<ORF>(v.w[2], vm.z, o), i3 = pass_bytes<ORF>(v.w[3], vm.w, o);
    uint32_t sq = __builtin_amdgcn_udot4(v.w[0], i0, 0u, false);      // += quality of every passing byte
    sq = __builtin_amdgcn_udot4(v.w[1], i1, sq, false);
    sq = __builtin_amdgcn_udot4(v.w[2], i2, sq, false);
    sq = __builtin_amdgcn_udot4(v.w[3], i3, sq, false);
    const uint32_t e0 = set_off + ((u << 1) | ((u >> 3) & 1u));
    atomicAdd(&s_qc[e0], ((unsigned long long)i1 << 32) | i0);
    atomicAdd(&s_qc[e0 ^ 1u], ((unsigned long long)i3 << 32) | i2);
    return sq;
}

// 16-bit counters: positions 4e..4e+3 are one 8-byte entry e = 4u + jj, stored at
// 4u + (jj ^ ((u>>2)&3)) so that lanes holding the same jj spread over all banks.
template <bool ORF>
__device__ __forceinline__ uint32_t apply_unit16(const Q16 &v, const uint4 vm, uint32_t u,
                                                 unsigned long long *__restrict__ s_qc, const Opts &o)
{
    uint32_t sq = 0;
    const uint32_t e0 = (u << 2) | ((u >> 2) & 3u);       // entry index for jj = 0, xor jj for the others
    const uint32_t vmw[4] = {vm.x, vm.y, vm.z, vm.w};
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const uint32_t xw = v.w[jj];
        const uint32_t inc = pass_bytes<ORF>(xw, vmw[jj], o);
        const uint32_t lo = __builtin_amdgcn_perm(0u, inc, 0x0c010c00u);   // bytes 0,1 -> 16-bit fields
        const uint32_t hi = __builtin_amdgcn_perm(0u, inc, 0x0c030c02u);   // bytes 2,3
        atomicAdd(&s_qc[e0 ^ (uint32_t)jj], ((unsigned long long)hi << 32) | lo);
        sq = __builtin_amdgcn_udot4(xw, inc, sq, false);
    }
    return sq;
}

template <bool ORF>
__device__ __forceinline__ uint32_t apply_unit32(const Q16 &v, const uint4 vm, uint32_t u,
                                                 uint32_t *__restrict__ s_qc, const Opts &o)
{
    uint32_t sq = 0;
    const uint32_t vmw[4] = {vm.x, vm.y, vm.z, vm.w};
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const uint32_t xw = v.w[jj];
        const uint32_t inc = pass_bytes<ORF>(xw, vmw[jj], o);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if ((inc >> (8 * i)) & 1u) atomicAdd(&s_qc[(u << 4) + 4 * jj + i], 1u);
        sq = __builtin_amdgcn_udot4(xw, inc, sq, false);
    }
    return sq;
}

// a list entry {quality offset, srel | (len-1)<<16 | set<<30 | valid<<31} as a lane quad sees it
struct SegView {
    uint32_t srel, trel, qoff, u1, ub, set;
    bool on;
};
__device__ __forceinline__ SegView seg_view(uint2 d, uint32_t ql)
{
    SegView s;
    s.on = (d.y >> 31) != 0u;
    s.set = (d.y >> 30) & 1u;
    s.srel = d.y & 0xFFFFu;
    s.trel = s.srel + ((d.y >> 16) & 0x3FFFu) + 1u;
    s.qoff = d.x + (uint32_t)kQualPad - s.srel;      // + 16*u = byte offset of unit u from the padded base
    s.u1 = (s.trel - 1u) >> 4;
    s.ub = (s.srel >> 4) + ql;
    return s;
}

#ifndef CL_MINWAVES
#define CL_MINWAVES 8
#endif
// timing experiments (tools/): phases of k_pileup are skipped by bits of PileupArgs::ablate -- compiled in only with
// -DCL_TUNING, never into the library the product loads (results are wrong with any bit set)
#ifdef CL_TUNING
#define CL_ABL(bit) ((a.ablate & (bit)) != 0u)
#else
#define CL_ABL(bit) false
#endif

template <int T, bool DEBUG, bool ORF, bool DEEP, int LONG>
__global__ __launch_bounds__(kBlock, DEEP ? 4 : CL_MINWAVES) void k_pileup(PileupArgs a)
{
    constexpr int PER = T / kBlock;                 // positions per thread in the final phase
    static_assert(PER == 8 || PER == 4, "T must be 2048 or 1024");
    constexpr int kWaves = kBlock / 64;
    static_assert(LONG == 0 || LONG == 2, "forms of k_pileup: 0 records (short reads), 2 run table (long reads)");
    constexpr int kListCap = LONG == 2 ? 1 : 64 + 16;   // entries of one wave's list: a pass's 64 + the < 16 carried over
    constexpr uint32_t kLutLds = 256;
    // +-1 differences of raw_depth / low_mapq_count.  DEEP: one 32-bit word per position.  Otherwise two
    // positions per word as 16-bit halves: the low half is biased by 0x8000 so that adding -1 (a
    // subtraction of 1 from the whole word) never borrows from the high half; exact while the window
    // is touched by < 32768 reads (host_window_bounds raises kNeedDeep beyond that).
    constexpr int kDiffWords = DEEP ? T : T / 2;
    __shared__ __attribute__((aligned(16))) uint32_t s_raw[kDiffWords];
    __shared__ __attribute__((aligned(16))) uint32_t s_low[kDiffWords];
    __shared__ __attribute__((aligned(16))) uint32_t s_qcw[DEEP ? T : T / 2];   // qc_depth counters
    __shared__ __attribute__((aligned(8))) uint2 s_list[kWaves][kListCap];
    __shared__ uint16_t s_lut[kLutLds];             // low-mapq threshold for raw < 256 (0xFFFF = never)
    // validity masks of a 16-position unit: byte i of s_mstart[vs] is 0x01 iff i >= vs,
    // byte i of s_mend[ve] is 0x01 iff i < ve (vs, ve in 0..16)
    __shared__ __attribute__((aligned(16))) uint4 s_mstart[17], s_mend[17];
    __shared__ uint32_t s_wraw[kWaves], s_wlow[kWaves], s_wmax[kWaves];
    __shared__ uint8_t s_last[kBlock];
    // per-wave totals: cnt[6], n_cov, sum_qc, sum_q, n_inner.  (Same-address LDS atomics are avoided:
    // hipcc turns them into a scalar loop over the active lanes.)
    __shared__ unsigned long long s_wtot[kWaves][12];          // [10], [11]: sums of the reads the window owns (LONG = 0)

    // XCD-aware window order: blocks b, b+8, b+16.. share an XCD (round-robin dispatch), give
    // each XCD one contiguous range of windows so neighbouring windows share its L2.
    const uint32_t w = (blockIdx.x & 7u) * a.n_win8 + (blockIdx.x >> 3);
    if (w >= a.n_win) return;
    const uint32_t tid = threadIdx.x;
    const uint32_t W = w * (uint32_t)T;
    const uint32_t Wend = W + (uint32_t)T;
    const uint32_t lane = tid & 63u, wv = tid >> 6;
    const uint32_t p0 = W + tid * PER;

    const WinMeta wm = a.win[w];
    const uint32_t lo = wm.lo, hi = wm.hi;
    // candidates: first the wn wide reads that start before read lo, then the reads [lo, hi)
    const uint32_t wlo = wm.wlo, wn = wm.wn;
    const uint32_t n_cand = wn + (hi - lo);
    // all quality bytes of the candidates lie within 2^32 of the first one's (checked by
    // host_window_bounds), so they are addressed by 32-bit offsets from a uniform base.  The base
    // sits kQualPad bytes low so that the offset of a unit start never goes negative.
    const unsigned long long qwin = wm.q0;
    const uint8_t *qbase = a.R.qual + qwin - kQualPad;

    // reference bytes of this thread's positions: needed last, requested first
    uint32_t refw[PER / 4];
#pragma unroll
    for (int i = 0; i < PER / 4; ++i) refw[i] = reinterpret_cast<const uint32_t *>(a.ref + p0)[i];

    // ---- clear ----
    {
        const uint4 z = make_uint4(0, 0, 0, 0);
        uint4 *r4 = reinterpret_cast<uint4 *>(s_raw), *l4 = reinterpret_cast<uint4 *>(s_low),
              *q4 = reinterpret_cast<uint4 *>(s_qcw);
        const uint4 zb = DEEP ? z : make_uint4(0x8000u, 0x8000u, 0x8000u, 0x8000u);
        for (int i = tid; i < kDiffWords / 4; i += kBlock) { r4[i] = zb; l4[i] = zb; }
        for (int i = tid; i < (DEEP ? T : T / 2) / 4; i += kBlock) q4[i] = z;
        if (tid < 34) {
            const uint32_t e = tid < 17 ? tid : tid - 17;                       // vs or ve
            const uint32_t bits = tid < 17 ? (0xFFFFu & ~((1u << e) - 1u)) : ((1u << e) - 1u);
            uint4 m;
            m.x = __umul24(bits & 15u, 0x204081u) & 0x01010101u;
            m.y = __umul24((bits >> 4) & 15u, 0x204081u) & 0x01010101u;
            m.z = __umul24((bits >> 8) & 15u, 0x204081u) & 0x01010101u;
            m.w = __umul24((bits >> 12) & 15u, 0x204081u) & 0x01010101u;
            if (tid < 17) s_mstart[e] = m; else s_mend[e] = m;
        }
        if (tid < kLutLds) {
            // fold "raw >= min_depth_for_low_mapq" into the table: below it the rule never fires
            const uint32_t v = (tid >= a.o.min_depth_for_low_mapq && tid > 0) ? a.lut[tid] : 0xFFFFFFFFu;
            s_lut[tid] = v > 0xFFFFu ? (uint16_t)0xFFFFu : (uint16_t)v;
        }
    }
    __syncthreads();

    // qc_depth counters: two sets of bytes when the window is touched by <= 510 reads (the reads
    // alternate between the sets, so no byte can pass 255), else 16-bit fields (DEEP: 32-bit words)
    // The bytes cannot overflow with <= 510 candidates.  With more (deeper data: 36x of 150-base reads
    // already has ~515 candidates per window) they still cannot while no position is covered by more than
    // 255 reads -- a byte counts reads of one set covering its position -- so the 8-bit sets are used
    // optimistically and the window's maximum raw depth, known in the final phase, is the check: beyond
    // 255 the kernel marks the window in win_wide, raises kNeedWide8, and the host runs the contig again:
    // marked windows then use the 16-bit fields.
    // (LONG = 2: the counter set of a piece is its read's parity in the contig, not in the window's candidate list, so the
    // candidate count bounds nothing and the maximum raw depth is the check for every window.)
    constexpr bool kByDepth = LONG == 2;
    const bool mode8 = !DEEP && ((!kByDepth && n_cand <= 510u) || a.win_wide[w] == 0);

    // ---- the pass over the reads ----
    uint32_t sq32 = 0;                              // sum of passing qualities handled by this lane
    unsigned long long sumq = 0;
    const uint32_t ql = lane & 3u, quad = lane >> 2;
    uint2 *list = s_list[wv];
    uint32_t n_keep = 0;                            // list entries carried over from the previous round (< 16)
    unsigned long long win_len = 0, win_mq = 0;     // wave-uniform sums over the reads this window owns
    // quads consume list entries [0, n_use), Q = n_use/16 (rounded up) entries each.  Three units per lane and
    // trip: u, u+4, u+8; a unit past the end is clamped onto the last one and gets an empty mask.
    // MODE 0: 8-bit two-set counters, 1: 16-bit fields, 2: 32-bit words (DEEP)
    // UPL units per lane and trip: 3 (12 unit slots per quad: fits a 150-base read) or 2 (8 slots: reads of
    // up to ~128 bases would leave a third of the 12 empty)
    auto consume = [&](auto mode_tag, auto upl_tag, uint32_t n_use) {
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr int UPL = decltype(upl_tag)::value;
        const uint32_t Q = (n_use + 15u) >> 4;
        for (uint32_t i = 0; i < Q; ++i) {
            // quad q takes entries q, q + 16, ...: the 16 quads of a trip work on 16 neighbouring segments, so the
            // 128-byte line that holds the end of one read's qualities and the start of the next is touched by two
            // quads of the same trip instead of microseconds apart (measured: 1 714 instead of 1 798 MB fetched per
            // launch, time equal within the run-to-run spread; CL_QUAD_BLOCKED: quad q takes entries q*Q .. q*Q+Q-1)
#ifdef CL_QUAD_BLOCKED
            const uint32_t idx = quad * Q + i;
#else
            const uint32_t idx = i * 16u + quad;
#endif
            uint2 d = make_uint2(0u, 0u);
            if (idx < n_use) d = list[idx];
            const SegView sv = seg_view(d, ql);
            for (uint32_t u = sv.ub; u <= sv.u1; u += 4u * UPL) {
                Q16 v[UPL];
                uint32_t uu[UPL];
#pragma unroll
                for (int j = 0; j < UPL; ++j) {
                    const uint32_t un = u + 4u * j;
                    uu[j] = un < sv.u1 ? un : sv.u1;
                    __builtin_memcpy(&v[j], qbase + (sv.qoff + (uu[j] << 4)), 16);
                }
#pragma unroll
                for (int j = 0; j < UPL; ++j) {
                    const uint32_t ps = uu[j] << 4;
                    const uint32_t vs = sv.srel > ps ? sv.srel - ps : 0u;
                    uint32_t ve = (sv.trel - ps) < 16u ? (sv.trel - ps) : 16u;
                    ve = (sv.on && u + 4u * j <= sv.u1) ? ve : 0u;
                    const uint4 ms = s_mstart[vs], me = s_mend[ve];
                    const uint4 vm = make_uint4(ms.x & me.x, ms.y & me.y, ms.z & me.z, ms.w & me.w);
                    // a slot past the segment's end adds zeros: to the word of its own unit number, not (with the lanes
                    // beside it) to the word of the segment's last unit
                    const uint32_t un = u + 4u * j, ua = un < (uint32_t)(T / 16) ? un : (uint32_t)(T / 16) - 1u;
                    if (MODE == 2) sq32 += apply_unit32<ORF>(v[j], vm, uu[j], s_qcw, a.o);
                    else if (MODE == 0) sq32 += apply_unit8<ORF>(v[j], vm, ua, sv.set * (uint32_t)(T / 8), reinterpret_cast<unsigned long long *>(s_qcw), a.o);
                    else sq32 += apply_unit16<ORF>(v[j], vm, uu[j], reinterpret_cast<unsigned long long *>(s_qcw), a.o);
                }
            }
        }
    };
    auto consume_list = [&](uint32_t n_use) {
        if (CL_ABL(1u)) return;
        using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
        if (DEEP) consume(std::integral_constant<int, 2>{}, I3{}, n_use);
        else if (mode8) { if (a.upl == 2u) consume(std::integral_constant<int, 0>{}, I2{}, n_use); else consume(std::integral_constant<int, 0>{}, I3{}, n_use); }
        else consume(std::integral_constant<int, 1>{}, I3{}, n_use);
    };
    if constexpr (LONG == 2) {
        // ---- run-table form.  (1) the window's candidates: +-1 at the clipped span ends, and the separable sums of the
        //      reads that start here (contig_profiler.rs:74) ----
        for (uint32_t base = 0; base < n_cand; base += kBlock) {
            const uint32_t v = base + tid;
            unsigned long long own_len = 0, own_mq = 0;
            if (v < n_cand) {
                const uint32_t r = v < wn ? a.wide_idx[wlo + v] : lo + (v - wn);
                const uint32_t x = (uint32_t)a.R.pos[r], e = a.end[r], mq = a.R.mapq[r];
                if (x >= W) {                                    // the window that holds the read's start owns its sums
                    own_len = e - x;
                    own_mq = mq >= a.o.min_mapq ? (unsigned long long)mq * (e - x) : 0ull;
                }
                if (e > W) {
                    const uint32_t cb = x > W ? x - W : 0u, ce = e - W;
                    uint32_t ib, vb, ie, ve2;
                    if (DEEP) { ib = cb; vb = 1u; ie = ce; ve2 = 0xFFFFFFFFu; }
                    else {
                        ib = cb >> 1; vb = (cb & 1u) ? 0x10000u : 1u;
                        ie = ce >> 1; ve2 = (ce & 1u) ? 0xFFFF0000u : 0xFFFFFFFFu;
                    }
                    atomicAdd(&s_raw[ib], vb);
                    if (ce < (uint32_t)T) atomicAdd(&s_raw[ie], ve2);
                    if (mq <= a.o.max_low_mapq) {
                        atomicAdd(&s_low[ib], vb);
                        if (ce < (uint32_t)T) atomicAdd(&s_low[ie], ve2);
                    }
                }
            }
            win_len += wave_sum_u64(own_len); win_mq += wave_sum_u64(own_mq);
        }
        // ---- (2) the window's pieces, streamed: entry {x, y}: x + 16 u = byte offset of unit u's qualities from qbase;
        //      y = start (11 bits) | end - 1 (11) | - | counter set (bit 29) | - | valid (bit 31).  A piece covers the
        //      unit of its start and at most the next one.  Lane t takes entries t, t + 256, ...: a wave's 64 entries are
        //      consecutive pieces of (mostly) one read, their quality bytes ~1 KB of one stretch of memory. ----
        {
#ifndef CL_RT_E
#define CL_RT_E 2
#endif
            constexpr int E = CL_RT_E;                           // entries per lane and trip: 2 E quality loads in flight
            const uint2 *ent = a.runtab + wm.rlo;
            const uint32_t nent = CL_ABL(2u) ? 0u : wm.rn;
            // (every load of the loop is unconditional -- an index past the end is clamped and its entry marked invalid --:
            // behind a load in a conditional block hipcc waits for vmcnt(0).  A clamped lane keeps the last entry's start
            // and end and only loses the valid bit: its two quality loads then go where that entry's go.  With the
            // whole word cleared they went to unit 0 of the window, up to 2 047 bytes in front of the entry's bytes -- in
            // front of the quality array itself when the window's last piece belongs to the contig's first read.)
            auto fetch = [&](uint32_t b, uint2 (&d)[E]) {
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    const uint32_t idx = b + (uint32_t)j * kBlock + tid;
                    const bool in = idx < nent;
                    d[j] = ent[in ? idx : nent - 1u];
                    d[j].y = in ? d[j].y : (d[j].y & 0x7FFFFFFFu);
                }
            };
            uint2 d[E], dn[E];
            if (nent) fetch(0u, d);
            for (uint32_t b = 0; b < nent; b += kBlock * E) {     // block-uniform
                fetch(b + kBlock * E, dn);                        // the next trip's entries are requested first
                Q16 v[E][2];
                uint32_t u0[E], two[E];
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    u0[j] = (d[j].y & 2047u) >> 4;
                    two[j] = (((d[j].y >> 11) & 2047u) >> 4) - u0[j];          // 0 or 1
                    if (CL_ABL(16u)) {                               // timing experiment: no quality bytes are loaded
                        v[j][0].w[0] = d[j].x; v[j][0].w[1] = d[j].y; v[j][0].w[2] = d[j].x ^ d[j].y; v[j][0].w[3] = u0[j];
                        v[j][1] = v[j][0];
                    } else {
                        __builtin_memcpy(&v[j][0], qbase + (d[j].x + (u0[j] << 4)), 16);
                        __builtin_memcpy(&v[j][1], qbase + (d[j].x + ((u0[j] + two[j]) << 4)), 16);
                    }
                }
                if (!CL_ABL(4u)) {
#pragma unroll
                    for (int j = 0; j < E; ++j) {
                        const uint32_t srel = d[j].y & 2047u, trel = ((d[j].y >> 11) & 2047u) + 1u;
                        const uint32_t set_off = ((d[j].y >> 29) & 1u) * (uint32_t)(T / 8);
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            if ((d[j].y >> 31) && (h == 0 || two[j])) {
                                const uint32_t u = u0[j] + (uint32_t)h, ps = u << 4;
                                const uint32_t vs = srel > ps ? srel - ps : 0u;
                                const uint32_t ve = (trel - ps) < 16u ? (trel - ps) : 16u;
                                const uint4 ms = s_mstart[vs], me = s_mend[ve];
                                const uint4 vm = make_uint4(ms.x & me.x, ms.y & me.y, ms.z & me.z, ms.w & me.w);
                                if (DEEP) sq32 += apply_unit32<ORF>(v[j][h], vm, u, s_qcw, a.o);
                                else if (mode8) sq32 += apply_unit8<ORF>(v[j][h], vm, u, set_off, reinterpret_cast<unsigned long long *>(s_qcw), a.o);
                                else sq32 += apply_unit16<ORF>(v[j][h], vm, u, reinterpret_cast<unsigned long long *>(s_qcw), a.o);
                            }
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < E; ++j) d[j] = dn[j];
                sumq += sq32; sq32 = 0;
            }
        }
    } else if constexpr (LONG == 0) {
        // ---- short-read form: the candidates are RECORDS (ReadRec), 256 at a time, one lane per record, the records
        //      dealt round-robin to the 4 waves; waves never synchronise during the pass.  No CIGAR is decoded on the device:
        //      the host's walk at upload turned every read into a head record (its span: the +-1 scatter, mod.rs:22-28, and
        //      the sums of the window that holds its start, contig_profiler.rs:74) that also carries the read's first
        //      M/=/X run when that starts at the read's position -- all there is to 96 reads in 100 of aligner output --
        //      and one piece record per further run (mod.rs:30-37 visits exactly those bases). ----
        for (uint32_t base = 0; base < (CL_ABL(2u) ? 0u : n_cand); base += kBlock) {
            const uint32_t v = base + 4u * lane + wv;   // candidate number; consecutive candidates alternate counter sets
            uint32_t r = lo + (v - wn);
            if (v < wn) r = a.wide_idx[wlo + v];
            __builtin_assume(r < (1u << 29));           // the host refuses contigs with >= 2^29 records
            uint2 seg = make_uint2(0u, 0u);
            uint32_t own_l = 0, own_m = 0;              // spans below 2^16: the wave's sums fit 32 bits
            bool big = false;                           // a head record with a wider span (rare: exact 64-bit sums below)
            uint32_t big_span = 0, big_mq = 0;
            if (v < n_cand) {
                uint4 rr = *reinterpret_cast<const uint4 *>(a.rec + r);
                // (one 16-byte load: left alone, hipcc splits it and fetches fields behind the tests that need them)
                asm volatile("" : "+v"(rr.x), "+v"(rr.y), "+v"(rr.z), "+v"(rr.w));
                const uint32_t x = rr.x, span = rr.y, mq = rr.w & 255u, seglen = rr.w >> 16;
                const bool hq = mq >= a.o.min_mapq;
                if ((rr.w & 0x100u) && span) {                   // head record: the read as the pileup holds it, [x, x + span)
                    const uint32_t e = x + span;
                    if (x >= W) {                                // every read starts in exactly one window
                        if (span < 0x10000u) { own_l = span; own_m = hq ? mq * span : 0u; }
                        else { big = true; big_span = span; big_mq = hq ? mq : 0u; }
                    }
                    if (e > W) {
                        const uint32_t cb = x > W ? x - W : 0u, ce = e - W;
                        uint32_t ib, vb, ie, ve2;            // word index and addend of the +1 and of the -1
                        if (DEEP) { ib = cb; vb = 1u; ie = ce; ve2 = 0xFFFFFFFFu; }
                        else {
                            ib = cb >> 1; vb = (cb & 1u) ? 0x10000u : 1u;
                            ie = ce >> 1; ve2 = (ce & 1u) ? 0xFFFF0000u : 0xFFFFFFFFu;
                        }
                        atomicAdd(&s_raw[ib], vb);
                        if (ce < (uint32_t)T) atomicAdd(&s_raw[ie], ve2);
                        if (mq <= a.o.max_low_mapq) {
                            atomicAdd(&s_low[ib], vb);
                            if (ce < (uint32_t)T) atomicAdd(&s_low[ie], ve2);
                        }
                    }
                }
                // the record's run of seglen bases with a quality byte each, from reference position x
                const uint32_t sp = x > W ? x : W, te = x + seglen, tp = te < Wend ? te : Wend;
                if (hq && seglen && sp < tp)
                    seg = make_uint2(rr.z - (uint32_t)qwin + (sp - x), (sp - W) | ((tp - sp - 1u) << 16) | ((v & 1u) << 30) | 0x80000000u);
            }
            win_len += dpp_wave_sum_u32(own_l); win_mq += dpp_wave_sum_u32(own_m);
            if (__any(big)) {
                win_len += wave_sum_u64(big ? (unsigned long long)big_span : 0ull);
                win_mq += wave_sum_u64(big ? (unsigned long long)big_mq * big_span : 0ull);
            }
            // -- wave-private list in lane (= position) order: carried-over entries, then this pass's --
            uint32_t n_list = n_keep;
            {
                const bool has = (seg.y >> 31) != 0u;
                const unsigned long long m = __ballot(has);
                if (has) {
                    const uint32_t idx = n_list + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    list[idx] = seg;
                }
                n_list += (uint32_t)__popcll(m);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // -- only full groups of 16 entries are consumed now; the < 16 left over move to the front
            //    of the list and wait for the next pass (or for the flush after the last one) --
            const uint32_t n_full = n_list & ~15u;
            if (n_full) consume_list(n_full);
            n_keep = n_list - n_full;
            uint2 carry = make_uint2(0u, 0u);
            if (n_full && lane < n_keep) carry = list[n_full + lane];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();       // the list is rewritten below and in the next pass
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (n_full && lane < n_keep) list[lane] = carry;
            sumq += sq32; sq32 = 0;
        }
    }
    if (n_keep) {                                   // flush what the last round left over
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        consume_list(n_keep);
        sumq += sq32; sq32 = 0;
    }
    __syncthreads();

    // ---- final phase: depths, low-MAPQ rule, state, counts (8 positions per thread) ----
    {
        uint32_t vr[PER], vl[PER];
        uint32_t sr = 0, sl = 0;
        if (DEEP) {
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                sr += s_raw[tid * PER + i]; vr[i] = sr;
                sl += s_low[tid * PER + i]; vl[i] = sl;
            }
        } else {
#pragma unroll
            for (int h = 0; h < PER / 2; ++h) {
                const uint32_t wr = s_raw[tid * (PER / 2) + h], wl = s_low[tid * (PER / 2) + h];
                // low half: biased by 0x8000; high half: two's complement 16-bit
                sr += (wr & 0xFFFFu) - 0x8000u; vr[2 * h] = sr;
                sr += (uint32_t)((int32_t)wr >> 16); vr[2 * h + 1] = sr;
                sl += (wl & 0xFFFFu) - 0x8000u; vl[2 * h] = sl;
                sl += (uint32_t)((int32_t)wl >> 16); vl[2 * h + 1] = sl;
            }
        }
        const uint32_t ir = dpp_incl_scan_u32(sr), il = dpp_incl_scan_u32(sl);
        if (lane == 63) { s_wraw[wv] = ir; s_wlow[wv] = il; }
        // 8-bit mode: the thread's PER positions are PER consecutive bytes of entry 2u+h (8 positions
        // each) of the two counter sets
        uint32_t qc8a[PER / 4], qc8b[PER / 4];
#pragma unroll
        for (int h = 0; h < PER / 4; ++h) { qc8a[h] = 0; qc8b[h] = 0; }
        if (!DEEP && mode8) {
            const uint32_t ent = (tid * PER) >> 3, u = ent >> 1;
            const uint32_t e = (u << 1) | ((ent & 1u) ^ ((u >> 3) & 1u));
#pragma unroll
            for (int h = 0; h < PER / 4; ++h) {
                const uint32_t wsel = PER == 8 ? (uint32_t)h : (tid & 1u);          // which half of the entry
                qc8a[h] = s_qcw[2u * e + wsel];
                qc8b[h] = s_qcw[2u * (T / 8 + e) + wsel];
            }
        }
        __syncthreads();
        uint32_t offr = ir - sr, offl = il - sl;
        for (uint32_t i = 0; i < wv; ++i) { offr += s_wraw[i]; offl += s_wlow[i]; }
        uint32_t mx = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) { vr[i] += offr; vl[i] += offl; mx = vr[i] > mx ? vr[i] : mx; }
        const uint32_t n_ok = p0 >= a.extent ? 0u : (a.extent - p0 < (uint32_t)PER ? a.extent - p0 : (uint32_t)PER);

        uint32_t S[PER / 4];                        // state bytes of the thread's positions
        uint32_t cnt[6] = {0, 0, 0, 0, 0, 0}, ncov = 0;
        unsigned long long sqc = 0;
        if (!DEEP && mode8 && mx < kLutLds && n_ok == (uint32_t)PER) {
            // ---- byte-parallel path: every column is shallower than 256, so qc_depth (<= raw_depth)
            //      fits a byte and four positions are classified per 32-bit word ----
            const uint32_t ONES = 0x01010101u;
#pragma unroll
            for (int h = 0; h < PER / 4; ++h) {
                const uint32_t q4 = qc8a[h] + qc8b[h];                       // no byte can carry
                uint32_t cov = 0, low = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t raw = vr[4 * h + i];
                    cov |= (raw < 1u ? raw : 1u) << (8 * i);
                    // low-MAPQ rule through the table (callable_profiler.rs:100-101)
                    low |= (vl[4 * h + i] >= (uint32_t)s_lut[raw] ? 1u : 0u) << (8 * i);
                }
                const uint32_t x = (refw[h] | 0x20202020u) ^ 0x6e6e6e6eu;   // zero byte <=> 'N' or 'n'
                const uint32_t nz = (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) >> 7;
                const uint32_t N = ~nz & ONES;
                uint32_t lt = ONES;                                          // qc < min_depth
                if (!a.o.md_all) lt = ~(swar_ge7(q4, a.o.md_add, a.o.md_or, a.o.md_and) >> 7) & ONES;
                uint32_t gt = 0;                                             // qc > max_depth (max_depth in 1..254)
                if (a.o.xd_on) gt = swar_ge7(q4, a.o.xd_add, a.o.xd_or, a.o.xd_and) >> 7;
                // priorities of callable_profiler.rs:104-116, resolved into disjoint flags
                const uint32_t t0 = ~N & cov;
                const uint32_t rLow = t0 & low, t1 = t0 & ~low;
                const uint32_t rLT = t1 & lt, t2 = t1 & ~lt;
                const uint32_t rGT = t2 & gt, rC = t2 & ~gt;
                const uint32_t rNC = ~N & ~cov & ONES;
                S[h] = rC + (rNC << 1) + rLT + (rLT << 1) + (rGT << 2) + rLow + (rLow << 2);
                cnt[0] += __popc(N); cnt[1] += __popc(rC); cnt[2] += __popc(rNC);
                cnt[3] += __popc(rLT); cnt[4] += __popc(rGT); cnt[5] += __popc(rLow);
                ncov += __popc(cov);
                sqc += __builtin_amdgcn_udot4(q4, ONES, 0u, false);
                if (DEBUG) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (a.dbg_raw) a.dbg_raw[p0 + 4 * h + i] = vr[4 * h + i];
                        if (a.dbg_low) a.dbg_low[p0 + 4 * h + i] = vl[4 * h + i];
                        if (a.dbg_qc) a.dbg_qc[p0 + 4 * h + i] = (q4 >> (8 * i)) & 0xFFu;
                    }
                }
            }
        } else {
            // ---- general path, one position at a time ----
            uint32_t qc[PER];
            if (DEEP) {
#pragma unroll
                for (int i = 0; i < PER; ++i) qc[i] = s_qcw[tid * PER + i];
            } else if (mode8) {
#pragma unroll
                for (int i = 0; i < PER; ++i)
                    qc[i] = ((qc8a[i >> 2] >> (8 * (i & 3))) & 0xFFu) + ((qc8b[i >> 2] >> (8 * (i & 3))) & 0xFFu);
            } else {
                const uint2 *q2 = reinterpret_cast<const uint2 *>(s_qcw);
#pragma unroll
                for (int h = 0; h < PER / 4; ++h) {
                    const uint32_t e = tid * (PER / 4) + h;                 // entry 4u+jj: u = e>>2, jj = e&3
                    const uint2 c = q2[(e & ~3u) | ((e & 3u) ^ ((e >> 4) & 3u))];
                    qc[4 * h + 0] = c.x & 0xFFFFu; qc[4 * h + 1] = c.x >> 16;
                    qc[4 * h + 2] = c.y & 0xFFFFu; qc[4 * h + 3] = c.y >> 16;
                }
            }
            uint32_t st[PER];
            for (int i = 0; i < PER; ++i) {
                const uint32_t raw = vr[i], low = vl[i];
                bool is_low = false;                                                  // callable_profiler.rs:100-101
                if (raw >= a.o.min_depth_for_low_mapq && raw > 0) {
                    if (raw < kLutSize) is_low = low >= a.lut[raw];
                    else is_low = ((double)low / (double)raw) > a.o.max_low_mapq_fraction;   // IEEE f64 divide
                }
                const uint32_t rb = (refw[i >> 2] >> (8 * (i & 3))) & 0xFFu;
                uint32_t sx = 1u;                                                     // CALLABLE
                sx = (a.o.max_depth > 0 && qc[i] > a.o.max_depth) ? 4u : sx;          // EXCESSIVE_COVERAGE
                sx = qc[i] < a.o.min_depth ? 3u : sx;                                 // LOW_COVERAGE
                sx = is_low ? 5u : sx;                                                // POOR_MAPPING_QUALITY
                sx = raw == 0 ? 2u : sx;                                              // NO_COVERAGE
                sx = ((rb | 0x20u) == 'n') ? 0u : sx;                                 // REF_N
                const bool ok = (uint32_t)i < n_ok;
                if (ok) { cnt[sx] += 1; ncov += raw > 0 ? 1u : 0u; sqc += qc[i]; }
                st[i] = ok ? sx : 0xFFu;
                if (DEBUG) {
                    if (a.dbg_raw) a.dbg_raw[p0 + i] = raw;
                    if (a.dbg_low) a.dbg_low[p0 + i] = low;
                    if (a.dbg_qc) a.dbg_qc[p0 + i] = qc[i];
                }
            }
#pragma unroll
            for (int h = 0; h < PER / 4; ++h)
                S[h] = st[4 * h] | (st[4 * h + 1] << 8) | (st[4 * h + 2] << 16) | (st[4 * h + 3] << 24);
        }
        // run boundaries strictly inside the window: position p (> W) whose state differs from p-1
        s_last[tid] = (uint8_t)(S[PER / 4 - 1] >> 24);
        mx = dpp_wave_max_u32(mx);
        if (lane == 0) s_wmax[wv] = mx;
        __syncthreads();
        uint32_t nb = 0;
        uint32_t bmk[PER / 4];                                               // 0x01 in the bytes that start a run
        {
            uint32_t prevb = tid > 0 ? (uint32_t)s_last[tid - 1] : (S[0] & 0xFFu);
            const uint4 okm = s_mend[n_ok];                                  // 0x01 for the positions < extent
            const uint32_t okw[2] = {okm.x, okm.y};
#pragma unroll
            for (int h = 0; h < PER / 4; ++h) {
                const uint32_t P = (S[h] << 8) | prevb;
                const uint32_t d = S[h] ^ P;
                bmk[h] = ((((d & 0x7f7f7f7fu) + 0x7f7f7f7fu) | d) >> 7) & okw[h];
                nb += __popc(bmk[h]);
                prevb = S[h] >> 24;
            }
        }
        if (DEBUG) {
#pragma unroll
            for (int h = 0; h < PER / 4; ++h) reinterpret_cast<uint32_t *>(a.state + p0)[h] = S[h];
        }
        if (mode8) {
            // <= 510 reads: a thread's counts are <= 8 and every wave total fits 10 bits (sum_qc 17,
            // sum_q 29): five packed words, one butterfly reduction each
            uint32_t pk[5];
            pk[0] = cnt[0] | (cnt[1] << 10) | (cnt[2] << 20);
            pk[1] = cnt[3] | (cnt[4] << 10) | (cnt[5] << 20);
            pk[2] = ncov | (nb << 10);
            pk[3] = (uint32_t)sqc;
            pk[4] = (uint32_t)sumq;
#pragma unroll
            for (int c = 0; c < 5; ++c) pk[c] = dpp_wave_sum_u32(pk[c]);
            if (lane == 0) {
                unsigned long long *t = s_wtot[wv];
                t[0] = pk[0] & 1023u; t[1] = (pk[0] >> 10) & 1023u; t[2] = pk[0] >> 20;
                t[3] = pk[1] & 1023u; t[4] = (pk[1] >> 10) & 1023u; t[5] = pk[1] >> 20;
                t[6] = pk[2] & 1023u; t[9] = pk[2] >> 10;
                t[7] = pk[3]; t[8] = pk[4];
                t[10] = win_len; t[11] = win_mq;
            }
        } else {
            // denser windows: totals may pass 2^32
            unsigned long long v[10];
#pragma unroll
            for (int c = 0; c < 6; ++c) v[c] = cnt[c];
            v[6] = ncov; v[7] = sqc; v[8] = sumq; v[9] = nb;
#pragma unroll
            for (int c = 0; c < 10; ++c) {
                const unsigned long long r = wave_sum_u64(v[c]);
                if (lane == 0) s_wtot[wv][c] = r;
            }
            if (lane == 0) { s_wtot[wv][10] = win_len; s_wtot[wv][11] = win_mq; }
        }
        __syncthreads();
        // the window's run list: every run start strictly inside the window, in position order
        // (k_rle_write turns the lists into intervals; the per-position states never reach HBM)
        {
            const uint32_t inc = dpp_incl_scan_u32(nb);
            if (nb) {
                uint32_t off = inc - nb;
                for (uint32_t i = 0; i < wv; ++i) off += (uint32_t)s_wtot[i][9];
                uint16_t *dst = a.runs + (size_t)w * T + off;
#pragma unroll
                for (int h = 0; h < PER / 4; ++h)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if ((bmk[h] >> (8 * j)) & 1u) *dst++ = (uint16_t)((tid * PER + 4 * h + j) | (((S[h] >> (8 * j)) & 7u) << 12));
            }
            if (tid == 0) a.first_state[w] = (uint8_t)(S[0] & 0xFFu);
            if (tid == kBlock - 1) a.last_state[w] = (uint8_t)(S[PER / 4 - 1] >> 24);
        }
    }
    if (tid == 0) {
        WinPartial wp;
        unsigned long long tot[12];
        for (int c = 0; c < 12; ++c) { tot[c] = 0; for (int i = 0; i < kWaves; ++i) tot[c] += s_wtot[i][c]; }
        for (int c = 0; c < 6; ++c) wp.cnt[c] = tot[c];
        wp.n_cov = tot[6]; wp.sum_qc = tot[7]; wp.sum_q = tot[8];
        wp.sum_reflen = tot[10]; wp.sum_mapq_reflen = tot[11];
        wp.n_inner = (uint32_t)tot[9];
        uint32_t m = 0;
        for (int i = 0; i < kWaves; ++i) m = s_wmax[i] > m ? s_wmax[i] : m;
        wp.max_raw = m;
        a.winpart[w] = wp;
        if (!DEEP && mode8 && (kByDepth || n_cand > 510u) && m > 255u) { a.win_wide[w] = 1; atomicOr(a.err_flag, kNeedWide8); }
    }
}

// ---------------------------------------------------------------------------------------------
// k_pileup_rows: the pass-bit form of k_pileup (the default; DUT_QUAL_FORM=bytes selects the byte forms above).
//
// "qual >= min_base_quality" (mod.rs:33) is decided once on the host, where the quality bytes are touched anyway
// (cl_push_reads: one bit per base, qual_pack.cpp), and the bits reach the device as ROWS (pass_rows.h): per window a
// stack of T-bit rows, bit p of a row <-> reference position W + p, every read of the window (mapq >= min) alone in its
// stretch of a row.  qc_depth[p] (mod.rs:30-37) is then the column sum of the window's rows -- taken BIT-SLICED: a lane
// owns a block of 32 positions, a wave streams whole groups of 4 rows (one 16-byte load per lane, 1 KB per wave
// instruction, no address arithmetic, no masks, no shifts), and adds them into NP counter planes (plane k = bit k of
// the 32 counts) with carry-save adders: three-input boolean operations (v_bitop3), about 4.5 instructions per row
// for 32 positions.  No LDS atomics, no CIGAR, no offsets.  The other waves' planes are added by wave 0 and compared --
// still bit-sliced -- with min_depth and max_depth (callable_profiler.rs:108-113): two 32-bit masks per block.
// quality_bases is the number of set bits (contig_profiler.rs:71: taken from the planes, sum of 2^p x popcount);
// summed_baseq comes with the bits from the host's walk (contig_profiler.rs:70, per-read separable: SURVEY 8a-7).
//
// The window's candidates are heads (8 bytes, one per read with a reference span; above): the +-1 scatter of raw_depth
// and low_mapq_count (mod.rs:22-28) into difference arrays in LDS, nothing else -- the reads' other separable sums
// (summed_coverage, summed_mapq) come from the host's walk too.
//
// The final phase works in the BIT DOMAIN: per position only the two tests that need the position's integers (raw_depth
// > 0; the low-MAPQ rule, callable_profiler.rs:100-101) are taken, each leaving one bit; from there a thread's PER
// positions are PER bits of a register -- the reference's N bits (one bit per position in HBM), the two compare masks,
// the priorities of callable_profiler.rs:104-116 as boolean operations on masks, the state as three bit planes, the
// state counts as popcounts, run boundaries as planes ^ (planes << 1 | previous state), the run list by a loop over the
// set bits of the boundary mask.
//
// NP: counter planes -- 8 while no window has more than 255 rows (63 groups), 16 up to 65 535, else 32.
// DEEP: 32-bit difference words (a window with more than 32 767 candidates), as in k_pileup.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t bs_maj(uint32_t a, uint32_t b, uint32_t c) { return (a & b) | (c & (a | b)); }

// 4 rows (one group) into the counter planes: two carry-save adders on plane 0, one on plane 1, a half-adder ripple above
template <int NP>
__device__ __forceinline__ void bs_add4(uint32_t (&c)[NP], const uint4 x)
{
    const uint32_t t0 = c[0] ^ x.x ^ x.y, k0 = bs_maj(c[0], x.x, x.y);
    c[0] = t0 ^ x.z ^ x.w;
    const uint32_t k1 = bs_maj(t0, x.z, x.w);
    uint32_t k = bs_maj(c[1], k0, k1);
    c[1] = c[1] ^ k0 ^ k1;
#pragma unroll
    for (int p = 2; p < NP; ++p) { const uint32_t t = c[p] & k; c[p] ^= k; k = t; }
}

// bit i = (the count of position i < K), for the NP planes of a block
template <int NP>
__device__ __forceinline__ uint32_t bs_less_than(const uint32_t (&c)[NP], unsigned long long K)
{
    if (NP < 64 && (K >> NP) != 0ull) return 0xFFFFFFFFu;       // K beyond what NP planes can count to
    uint32_t lt = 0u, eq = 0xFFFFFFFFu;
#pragma unroll
    for (int p = NP - 1; p >= 0; --p) {
        if ((K >> p) & 1ull) { lt |= eq & ~c[p]; eq &= c[p]; }
        else eq &= ~c[p];
    }
    return lt;
}

// BS: threads per workgroup -- 256 (8 positions per thread in the final phase, 4 waves) or 128 (16 positions, 2 waves:
// what a wave does once per window -- scans, reductions, the planes' hand-over -- is done half as often)
#ifndef CL_ROWS_BLOCK
#define CL_ROWS_BLOCK 128
#endif
template <int T, bool DEBUG, bool DEEP, int NP, int BS>
__global__ __launch_bounds__(BS, (DEEP || NP > 8) ? (BS == 128 ? 3 : 4) : (BS == 128 ? 6 : 8)) void k_pileup_rows(PileupArgs a)
{
    constexpr int kBlock = BS;                             // (shadows the namespace's 256 inside this kernel)
    constexpr int PER = T / kBlock;
    static_assert((PER == 8 || PER == 16) && T == 2048, "a lane owns a block of 32 positions: T = 64 x 32");
    constexpr int kWaves = kBlock / 64;
    constexpr int kDiffWords = DEEP ? T : T / 2;
    constexpr int G = BS == 128 ? 6 : 4;                   // groups a wave has in flight: 12 / 16 per window before a second trip
    __shared__ __attribute__((aligned(16))) uint32_t s_raw[kDiffWords];
    __shared__ __attribute__((aligned(16))) uint32_t s_low[kDiffWords];
    // LDS: the two difference arrays and ONE pool that is used twice -- 10 240 bytes in all, so that a CU holds 16
    // workgroups (the kernel's time follows the number of workgroups a CU runs: profiles/r04_occupancy.txt):
    //   first   the counter planes of waves 1.. (wave 0 adds them to its own after the barrier; it is their only reader)
    //   then    s_lt / s_gt and the low-MAPQ thresholds s_lut: a lane of wave 0 writes its words after it has read its
    //           planes (they lie in the slots of that lane's own planes 0, 1 and kLutPlane of wave 1); s_last, s_wtot, s_wmax:
    //           written behind the NEXT barrier, when wave 0 is long done with the planes
    // (the waves' totals for the prefix sums across waves travel in the difference arrays: a lane's own, consumed slot)
    constexpr int kPoolWords = (kWaves - 1) * NP * 64;
    constexpr int kTenantsEnd = 128 + kBlock / 4 + kWaves * 24 + kWaves;        // words: s_lt, s_gt, s_last, s_wtot, s_wmax
    constexpr int kLutPlane = (kTenantsEnd + 63) / 64;                          // the thresholds: the first whole plane behind them
    static_assert(NP >= 8 && kLutPlane < NP && kPoolWords >= (kLutPlane + 1) * 64, "the pool holds its second tenants");
    __shared__ __attribute__((aligned(16))) uint32_t s_pool[kPoolWords];
    uint32_t (*s_pl)[NP][64] = reinterpret_cast<uint32_t (*)[NP][64]>(s_pool);
    uint32_t *const s_lt = s_pool, *const s_gt = s_pool + 64;                  // per block: qc < min_depth, qc > max_depth
    uint8_t *const s_last = reinterpret_cast<uint8_t *>(s_pool + 128);         // kBlock bytes
    unsigned long long (*s_wtot)[12] = reinterpret_cast<unsigned long long (*)[12]>(s_pool + 128 + kBlock / 4);
    uint32_t *const s_wmax = s_pool + 128 + kBlock / 4 + kWaves * 24;
    // the low-MAPQ thresholds of depths below 255 as bytes: 255 = never (a count is at most the depth)
    uint8_t *const s_lut = reinterpret_cast<uint8_t *>(s_pool + kLutPlane * 64);
    __shared__ uint32_t s_dbg[DEBUG ? NP : 1][64];         // DEBUG: the window's planes, for the dump of qc_depth
#ifdef CL_ROWS_LDS_PAD
    __shared__ uint32_t s_pad[CL_ROWS_LDS_PAD / 4];        // (occupancy experiments only)
    s_pad[threadIdx.x] = threadIdx.x;
#endif

    const uint32_t w = (blockIdx.x & 7u) * a.n_win8 + (blockIdx.x >> 3);   // XCD-contiguous window ranges
    if (w >= a.n_win) return;
    const uint32_t tid = threadIdx.x;
    const uint32_t W = w * (uint32_t)T;
    // (the wave number from a scalar register: whatever is indexed or bounded by it below is scalar code)
    const uint32_t lane = tid & 63u, wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t p0 = W + tid * PER;

    const WinMeta wm = a.win[w];
    const uint32_t lo = wm.lo, hi = wm.hi, wlo = wm.wlo, wn = wm.wn;
    const uint32_t n_cand = wn + (hi - lo);
    const uint32_t ng = wm.rn;                             // groups of 4 rows
    const uint4 *rows = a.rows + (size_t)wm.rlo * 64u;

    // requested first, needed last: the window's rows (this wave's first G groups), the reference bytes
    // (every load unconditional: a group past the end is clamped onto the last one and zeroed)
    uint4 rv[G];
    if (ng) {
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const uint32_t g = wv + (uint32_t)kWaves * j;
            rv[j] = rows[(size_t)(g < ng ? g : ng - 1u) * 64u + lane];
        }
    }
    // ... and the window's first candidates: heads (kernels.hip.h: HeadRec), U per lane and trip
    constexpr int U = 4;
    auto load_heads = [&](uint32_t base, uint2 (&hh)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t v = base + (uint32_t)u * kBlock + tid;
            uint32_t r = lo + (v - wn);
            if (v < wn) r = a.wide_idx[wlo + v];
            __builtin_assume(r < (1u << 29));
            hh[u] = make_uint2(0u, 0u);
            if (v < n_cand) hh[u] = a.heads[r];
        }
    };
    uint2 hh[U];
    load_heads(0u, hh);
    // (bit p of refn: the reference base at p is 'N' / 'n' or lies beyond the reference, mod.rs:79-80, :100-101)
    const uint32_t refn = PER == 16 ? (uint32_t)reinterpret_cast<const uint16_t *>(a.refn)[(size_t)w * (T / 16) + tid]
                                    : (uint32_t)reinterpret_cast<const uint8_t *>(a.refn)[(size_t)w * (T / 8) + tid];

    // ---- clear ----
    {
        const uint4 z = make_uint4(0, 0, 0, 0);
        uint4 *r4 = reinterpret_cast<uint4 *>(s_raw), *l4 = reinterpret_cast<uint4 *>(s_low);
        const uint4 zb = DEEP ? z : make_uint4(0x8000u, 0x8000u, 0x8000u, 0x8000u);
        for (int i = tid; i < kDiffWords / 4; i += kBlock) { r4[i] = zb; l4[i] = zb; }
    }
    __syncthreads();

    // ---- the window's rows: this wave's groups wv, wv + 4, ... into its counter planes.  The wave number is taken
    //      from a scalar register so that the tests on group numbers are scalar branches: a group slot past the window's
    //      last group costs nothing (the kernel is bound by vector issue), and the loads of a next trip are only issued
    //      when there is one (more than 16 groups: depth beyond 64) ----
    uint32_t c[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) c[p] = 0u;
    if (ng) {
        for (uint32_t g0 = wv; g0 < ng; g0 += (uint32_t)kWaves * G) {
#pragma unroll
            for (int j = 0; j < G; ++j) {
                if (g0 + (uint32_t)kWaves * j < ng) bs_add4<NP>(c, rv[j]);
            }
            if (g0 + (uint32_t)kWaves * G < ng) {          // a deeper window: the next trip's groups (requested only now)
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    const uint32_t g = g0 + (uint32_t)kWaves * (G + j);
                    rv[j] = rows[(size_t)(g < ng ? g : ng - 1u) * 64u + lane];
                }
            }
        }
    }
    // ---- the window's candidates.  +-1 at the clipped span ends (mod.rs:22-28: every read covering a position counts,
    //      D/N included).  The first trip's heads were requested at the top and have arrived behind the rows ----
    for (uint32_t base = 0;;) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t x = hh[u].x, span = hh[u].y & kHeadSpanMax;
            const uint32_t e = x + span;
            // (a head of a cut span may lie past the window; span = 0: no candidate in this slot)
            if (span && e > W && x < W + (uint32_t)T) {
                const uint32_t cb = x > W ? x - W : 0u, ce = e - W;
                uint32_t ib, vb, ie, ve2;
                if (DEEP) { ib = cb; vb = 1u; ie = ce; ve2 = 0xFFFFFFFFu; }
                else {
                    ib = cb >> 1; vb = (cb & 1u) ? 0x10000u : 1u;
                    ie = ce >> 1; ve2 = (ce & 1u) ? 0xFFFF0000u : 0xFFFFFFFFu;
                }
                atomicAdd(&s_raw[ib], vb);
                if (ce < (uint32_t)T) atomicAdd(&s_raw[ie], ve2);
                if (hh[u].y >> 31) {
                    atomicAdd(&s_low[ib], vb);
                    if (ce < (uint32_t)T) atomicAdd(&s_low[ie], ve2);
                }
            }
        }
        base += (uint32_t)U * kBlock;
        if (base >= n_cand) break;
        load_heads(base, hh);
    }

    if (wv != 0) {
#pragma unroll
        for (int p = 0; p < NP; ++p) s_pl[wv - 1][p][lane] = c[p];
    }
    __syncthreads();

    // ---- final phase: depths, low-MAPQ rule, state, counts (8 positions per thread) ----
    {
        // wave 0: the four waves' planes added (a bit-sliced ripple adder per wave) and compared with the two depth
        // thresholds (callable_profiler.rs:108-113); the other waves go on with their prefix sums meanwhile
        unsigned long long nbits = 0;                      // set bits of the window's rows, by wave 0's lanes (-> quality_bases)
        if (wv == 0) {
#pragma unroll
            for (int v = 1; v < kWaves; ++v) {
                uint32_t carry = 0u;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const uint32_t d = s_pl[v - 1][p][lane];
                    const uint32_t s = c[p] ^ d ^ carry;
                    carry = bs_maj(c[p], d, carry);
                    c[p] = s;
                }
            }
            // quality_bases (contig_profiler.rs:71): the sum of the block's 32 counts = sum over the planes of 2^p x set bits
#pragma unroll
            for (int p = 0; p < NP; ++p) nbits += (unsigned long long)__popc(c[p]) << p;
            s_lt[lane] = bs_less_than<NP>(c, (unsigned long long)a.o.min_depth);
            // qc > max_depth  <=>  !(qc < max_depth + 1); the rule is off for max_depth == 0
            s_gt[lane] = a.o.max_depth > 0u ? ~bs_less_than<NP>(c, (unsigned long long)a.o.max_depth + 1ull) : 0u;
            if (DEBUG) {
#pragma unroll
                for (int p = 0; p < NP; ++p) s_dbg[p][lane] = c[p];
            }
            // the thresholds of depths 4 lane .. 4 lane + 3, four bytes in the lane's own word
            {
                const uint4 lv = reinterpret_cast<const uint4 *>(a.lut)[lane];
                const uint32_t l4[4] = {lv.x, lv.y, lv.z, lv.w};
                uint32_t wlut = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t i = 4u * lane + (uint32_t)k;
                    const uint32_t v = (i >= a.o.min_depth_for_low_mapq && i > 0) ? l4[k] : 0xFFFFFFFFu;
                    wlut |= (v > 254u ? 255u : v) << (8 * k);
                }
                s_pool[kLutPlane * 64 + lane] = wlut;
            }
        }
        uint32_t vr[PER], vl[PER];
        uint32_t sr = 0, sl = 0;
        if (DEEP) {
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                sr += s_raw[tid * PER + i]; vr[i] = sr;
                sl += s_low[tid * PER + i]; vl[i] = sl;
            }
        } else {
#pragma unroll
            for (int h = 0; h < PER / 2; ++h) {
                const uint32_t wr = s_raw[tid * (PER / 2) + h], wl = s_low[tid * (PER / 2) + h];
                sr += (wr & 0xFFFFu) - 0x8000u; vr[2 * h] = sr;
                sr += (uint32_t)((int32_t)wr >> 16); vr[2 * h + 1] = sr;
                sl += (wl & 0xFFFFu) - 0x8000u; vl[2 * h] = sl;
                sl += (uint32_t)((int32_t)wl >> 16); vl[2 * h + 1] = sl;
            }
        }
        // (a thread's sum of differences may be negative: two's complement in 32 bits, so the two scans stay separate)
        const uint32_t ir = dpp_incl_scan_u32(sr), il = dpp_incl_scan_u32(sl);
        // the wave's totals, for the waves behind it: in the first slot this lane has just consumed
        constexpr int kSlot = DEEP ? PER : PER / 2;
        if (lane == 63) { s_raw[tid * kSlot] = ir; s_low[tid * kSlot] = il; }
        __syncthreads();
        uint32_t offr = ir - sr, offl = il - sl;
        for (uint32_t i = 0; i < wv; ++i) { offr += s_raw[(i * 64u + 63u) * kSlot]; offl += s_low[(i * 64u + 63u) * kSlot]; }
        uint32_t mx = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) { vr[i] += offr; vl[i] += offl; mx = vr[i] > mx ? vr[i] : mx; }
        const uint32_t n_ok = p0 >= a.extent ? 0u : (a.extent - p0 < (uint32_t)PER ? a.extent - p0 : (uint32_t)PER);
        // From here on the thread's PER positions are PER bits of a register: bit i <-> position p0 + i.
        constexpr uint32_t FULL = PER == 16 ? 0xFFFFu : 0xFFu;
        const uint32_t okb = n_ok >= (uint32_t)PER ? FULL : ((1u << n_ok) - 1u);            // positions < extent
        // qc_depth < min_depth, qc_depth > max_depth: PER consecutive bits of block (tid * PER) >> 5
        const uint32_t ltb = (s_lt[(tid * PER) >> 5] >> ((tid * PER) & 31u)) & FULL;
        const uint32_t gtb = (s_gt[(tid * PER) >> 5] >> ((tid * PER) & 31u)) & FULL;
        // raw_depth > 0, and the low-MAPQ rule (callable_profiler.rs:100-101): the two per-position tests
        uint32_t covb = 0, lowb = 0;
        if (!DEEP && mx < 255u) {
            // two instructions per test: a difference whose sign bit says "no" (raw - 1 wraps when raw == 0; low - lut
            // is negative when low < lut: all are below 2^31), shifted into the mask by v_alignbit ({mask, d} >> 31)
            uint32_t ncov = 0, nlow = 0;
#pragma unroll
            for (int i = PER - 1; i >= 0; --i) {
                const uint32_t raw = vr[i];
                ncov = __builtin_amdgcn_alignbit(ncov, raw - 1u, 31);
                nlow = __builtin_amdgcn_alignbit(nlow, vl[i] - (uint32_t)s_lut[raw], 31);
            }
            covb = ~ncov & FULL; lowb = ~nlow & FULL;
        } else {
#pragma unroll
            for (int i = PER - 1; i >= 0; --i) {
                const uint32_t raw = vr[i], low = vl[i];
                bool is_low = false;
                if (raw >= a.o.min_depth_for_low_mapq && raw > 0) {
                    if (raw < kLutSize) is_low = low >= a.lut[raw];
                    else is_low = ((double)low / (double)raw) > a.o.max_low_mapq_fraction;   // IEEE f64 divide
                }
                covb = covb + covb + (raw > 0 ? 1u : 0u);
                lowb = lowb + lowb + (is_low ? 1u : 0u);
            }
        }
        // priorities of callable_profiler.rs:104-116, resolved into disjoint masks:
        // REF_N > NO_COVERAGE > POOR_MAPPING_QUALITY > LOW_COVERAGE > EXCESSIVE_COVERAGE > CALLABLE
        const uint32_t Nk = refn & okb, notN = ~refn & okb, covk = covb & okb;
        const uint32_t t0 = notN & covk;
        const uint32_t rLow = t0 & lowb, t1 = t0 & ~lowb;
        const uint32_t rLT = t1 & ltb, t2 = t1 & ~ltb;
        const uint32_t rGT = t2 & gtb, rC = t2 & ~gtb;
        const uint32_t rNC = notN & ~covk;
        uint32_t cnt[6];
        cnt[0] = __popc(Nk); cnt[1] = __popc(rC); cnt[2] = __popc(rNC);
        cnt[3] = __popc(rLT); cnt[4] = __popc(rGT); cnt[5] = __popc(rLow);
        const uint32_t ncov = __popc(covk);
        // the state (types.rs:36-43: REF_N 0, CALLABLE 1, NO_COVERAGE 2, LOW_COVERAGE 3, EXCESSIVE_COVERAGE 4,
        // POOR_MAPPING_QUALITY 5) as three bit planes
        const uint32_t s0 = rC | rLT | rLow, s1 = rNC | rLT, s2 = rGT | rLow;
        auto state_at = [&](uint32_t j) -> uint32_t { return ((s0 >> j) & 1u) | (((s1 >> j) & 1u) << 1) | (((s2 >> j) & 1u) << 2); };
        if (DEBUG) {
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                uint32_t qc = 0;
                const uint32_t bit = ((tid * PER) & 31u) + (uint32_t)i;
#pragma unroll
                for (int p = 0; p < NP; ++p) qc |= ((s_dbg[p][(tid * PER) >> 5] >> bit) & 1u) << p;
                if (a.dbg_raw) a.dbg_raw[p0 + i] = vr[i];
                if (a.dbg_low) a.dbg_low[p0 + i] = vl[i];
                if (a.dbg_qc) a.dbg_qc[p0 + i] = qc;
                a.state[p0 + i] = (uint8_t)(((okb >> i) & 1u) ? state_at((uint32_t)i) : 0xFFu);
            }
        }
        // run boundaries strictly inside the window: position p (> W) whose state differs from p-1
        const uint32_t last_st = state_at((uint32_t)PER - 1u);
        s_last[tid] = (uint8_t)last_st;
        mx = dpp_wave_max_u32(mx);
        if (lane == 0) s_wmax[wv] = mx;
        __syncthreads();
        const uint32_t pv = tid > 0 ? (uint32_t)s_last[tid - 1] : state_at(0u);
        const uint32_t bnd = ((s0 ^ ((s0 << 1) | (pv & 1u))) | (s1 ^ ((s1 << 1) | ((pv >> 1) & 1u))) | (s2 ^ ((s2 << 1) | (pv >> 2)))) & okb;
        const uint32_t nb = __popc(bnd);
        if (!DEEP && NP == 8) {
            // a thread's counts are <= PER and every wave total <= 64 PER: packed words, one butterfly reduction each --
            // three 10-bit fields per word for 8 positions per thread (totals <= 512), two 11-bit fields for 16 (<= 1024);
            // the window's set bits (all with wave 0) are <= 255 x 2048 < 2^19
            constexpr int NW = PER == 8 ? 4 : 5;
            uint32_t pk[NW];
            if (PER == 8) {
                pk[0] = cnt[0] | (cnt[1] << 10) | (cnt[2] << 20);
                pk[1] = cnt[3] | (cnt[4] << 10) | (cnt[5] << 20);
                pk[2] = ncov | (nb << 10);
                pk[3] = (uint32_t)nbits;
            } else {
                pk[0] = cnt[0] | (cnt[1] << 11);
                pk[1] = cnt[2] | (cnt[3] << 11);
                pk[2] = cnt[4] | (cnt[5] << 11);
                pk[3] = ncov | (nb << 11);
                pk[NW - 1] = (uint32_t)nbits;
            }
#pragma unroll
            for (int q = 0; q < NW; ++q) pk[q] = dpp_wave_sum_u32(pk[q]);
            if (lane == 0) {
                unsigned long long *t = s_wtot[wv];
                if (PER == 8) {
                    t[0] = pk[0] & 1023u; t[1] = (pk[0] >> 10) & 1023u; t[2] = pk[0] >> 20;
                    t[3] = pk[1] & 1023u; t[4] = (pk[1] >> 10) & 1023u; t[5] = pk[1] >> 20;
                    t[6] = pk[2] & 1023u; t[9] = pk[2] >> 10;
                    t[7] = pk[3];
                } else {
                    t[0] = pk[0] & 2047u; t[1] = pk[0] >> 11;
                    t[2] = pk[1] & 2047u; t[3] = pk[1] >> 11;
                    t[4] = pk[2] & 2047u; t[5] = pk[2] >> 11;
                    t[6] = pk[3] & 2047u; t[9] = pk[3] >> 11;
                    t[7] = pk[NW - 1];
                }
                t[8] = 0;
                t[10] = 0; t[11] = 0;                    // (the reads' separable sums come from the host's walk)
            }
        } else {
            unsigned long long v[10];
#pragma unroll
            for (int q = 0; q < 6; ++q) v[q] = cnt[q];
            v[6] = ncov; v[7] = nbits; v[8] = 0; v[9] = nb;
#pragma unroll
            for (int q = 0; q < 10; ++q) {
                const unsigned long long r = wave_sum_u64(v[q]);
                if (lane == 0) s_wtot[wv][q] = r;
            }
            if (lane == 0) { s_wtot[wv][10] = 0; s_wtot[wv][11] = 0; }
        }
        __syncthreads();
        {
            const uint32_t inc = dpp_incl_scan_u32(nb);
            if (nb) {
                uint32_t off = inc - nb;
                for (uint32_t i = 0; i < wv; ++i) off += (uint32_t)s_wtot[i][9];
                uint16_t *dst = a.runs + (size_t)w * T + off;
                for (uint32_t m = bnd; m; m &= m - 1u) {     // (a lane has a boundary or two, rarely more)
                    const uint32_t j = (uint32_t)__ffs((int)m) - 1u;
                    *dst++ = (uint16_t)((tid * PER + j) | (state_at(j) << 12));
                }
            }
            if (tid == 0) a.first_state[w] = (uint8_t)state_at(0u);
            if (tid == kBlock - 1) a.last_state[w] = (uint8_t)last_st;
        }
    }
    if (tid == 0) {
        WinPartial wp;
        unsigned long long tot[12];
        for (int q = 0; q < 12; ++q) { tot[q] = 0; for (int i = 0; i < kWaves; ++i) tot[q] += s_wtot[i][q]; }
        for (int q = 0; q < 6; ++q) wp.cnt[q] = tot[q];
        wp.n_cov = tot[6]; wp.sum_qc = tot[7]; wp.sum_q = tot[8];
        wp.sum_reflen = tot[10]; wp.sum_mapq_reflen = tot[11];
        wp.n_inner = (uint32_t)tot[9];
        uint32_t m = 0;
        for (int i = 0; i < kWaves; ++i) m = s_wmax[i] > m ? s_wmax[i] : m;
        wp.max_raw = m;
        a.winpart[w] = wp;
    }
}

// ---------------------------------------------------------------------------------------------
// k_fin_windows / fin_summary: exclusive scan of the run starts per window inside blocks of kFinBlock
// windows (inner boundaries + the seam with the previous window) and reduction of the window
// partials to the contig summary (fin_summary runs as the last workgroup of k_rle_write).
// ---------------------------------------------------------------------------------------------
constexpr int kFinBlock = 1024;

struct FinPartial {
    unsigned long long acc[11];      // cnt[6], n_cov, sum_qc, sum_q, sum_reflen, sum_mapq_reflen
    uint32_t n_runs;
    uint32_t max_raw;
};

__global__ __launch_bounds__(kFinBlock) void k_fin_windows(const WinPartial *__restrict__ winpart,
                                                            const uint8_t *__restrict__ first_state,
                                                            const uint8_t *__restrict__ last_state, uint32_t T,
                                                            uint32_t n_win, uint32_t extent,
                                                            uint32_t *__restrict__ win_off,
                                                            FinPartial *__restrict__ fin)
{
    __shared__ uint32_t s_w[kFinBlock / 64], s_m[kFinBlock / 64];
    __shared__ unsigned long long s_red[11][kFinBlock / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t w = blockIdx.x * kFinBlock + tid;
    unsigned long long acc[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t c = 0, maxraw = 0;
    if (w < n_win) {
        const WinPartial wp = winpart[w];
        c = wp.n_inner;
        const uint32_t p = w * T;
        if (p < extent) c += (w == 0) ? 1u : (first_state[w] != last_state[w - 1] ? 1u : 0u);
        for (int i = 0; i < 6; ++i) acc[i] = wp.cnt[i];
        acc[6] = wp.n_cov; acc[7] = wp.sum_qc; acc[8] = wp.sum_q;
        acc[9] = wp.sum_reflen; acc[10] = wp.sum_mapq_reflen;
        maxraw = wp.max_raw;
    }
    const uint32_t inc = dpp_incl_scan_u32(c);
    if (lane == 63) s_w[wv] = inc;
#pragma unroll
    for (int i = 0; i < 11; ++i) {
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
        for (int i = 0; i < 11; ++i) { unsigned long long v = 0; for (int j = 0; j < kFinBlock / 64; ++j) v += s_red[i][j]; fp.acc[i] = v; }
        fin[blockIdx.x] = fp;
    }
}

// The contig summary: reduction of the per-block window partials.  Run by
// one workgroup of kBlock threads (the extra, last workgroup of k_rle_write).
__device__ __forceinline__ void fin_summary(const FinPartial *__restrict__ fin, uint32_t n_fin,
                                            uint32_t extent, uint32_t *__restrict__ err_flag,
                                            DevSummary *__restrict__ out, unsigned long long host_sum_q,
                                            unsigned long long host_sum_cov, unsigned long long host_sum_mapq)
{
    __shared__ unsigned long long s_red[12][kBlock / 64];
    __shared__ uint32_t s_u[kBlock / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    unsigned long long acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // [11]: number of runs
    uint32_t maxraw = 0;
    for (uint32_t b = tid; b < n_fin; b += kBlock) {
        const FinPartial fp = fin[b];
        acc[11] += fp.n_runs;
        for (int i = 0; i < 11; ++i) acc[i] += fp.acc[i];        // [9], [10]: over the reads each window owns
        maxraw = fp.max_raw > maxraw ? fp.max_raw : maxraw;
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const unsigned long long v = wave_sum_u64(acc[i]);
        if (lane == 0) s_red[i][wv] = v;
    }
    maxraw = wave_max_u32(maxraw);
    if (lane == 0) s_u[wv] = maxraw;
    __syncthreads();
    if (tid == 0) {
        unsigned long long tot[12];
        for (int i = 0; i < 12; ++i) { tot[i] = 0; for (int j = 0; j < kBlock / 64; ++j) tot[i] += s_red[i][j]; }
        uint32_t mr = 0;
        for (int j = 0; j < kBlock / 64; ++j) mr = s_u[j] > mr ? s_u[j] : mr;
        for (int i = 0; i < 6; ++i) out->state_counts[i] = tot[i];
        out->n_covered_bases = tot[6];
        out->quality_bases = tot[7];
        // (pass-bit form: the kernels see bits, the sum of the passing qualities comes from the host's walk over the bytes)
        out->summed_baseq = tot[8] + host_sum_q;
        // (pass-bit form: so do the reads' reference spans and mapq x span, contig_profiler.rs:74, 79-82)
        out->summed_coverage = tot[9] + host_sum_cov;
        out->summed_mapq = tot[10] + host_sum_mapq;
        out->extent = extent;
        out->max_raw_depth = mr;
        out->n_intervals = tot[11];
        out->max_end = 0;                    // read ends come from the host's walk (cl_push_reads)
        out->err = err_flag[0];
        // every kernel of the run is done with the flags: clear them for the next run (saves a memset launch)
        err_flag[0] = 0; err_flag[1] = 0;
    }
}

// ---------------------------------------------------------------------------------------------
// k_rle_write: one wave per window turns the window's run list into intervals.  The lane that writes
// the start of run i also closes run i-1.  The extra last workgroup computes the contig summary.
// ---------------------------------------------------------------------------------------------
template <int T>
__global__ __launch_bounds__(kBlock) void k_rle_write(const uint16_t *__restrict__ runs,
                                                       const uint8_t *__restrict__ first_state,
                                                       const uint8_t *__restrict__ last_state,
                                                       const WinPartial *__restrict__ winpart,
                                                       const uint32_t *__restrict__ win_off,
                                                       const FinPartial *__restrict__ fin, uint32_t n_fin,
                                                       uint32_t *__restrict__ err_flag,
                                                       DevSummary *__restrict__ summary,
                                                       uint32_t n_win, uint32_t extent,
                                                       Interval *__restrict__ iv, uint32_t iv_cap,
                                                       unsigned long long host_sum_q, unsigned long long host_sum_cov,
                                                       unsigned long long host_sum_mapq)
{
    if (blockIdx.x == gridDim.x - 1) {                     // the extra workgroup: the contig summary
        fin_summary(fin, n_fin, extent, err_flag, summary, host_sum_q, host_sum_cov, host_sum_mapq);
        return;
    }
    // one wave per window: its seam run (if the first state differs from the previous window's last)
    // and the run starts of its list; a run start also closes the run before it
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (w >= n_win) return;
    const uint32_t W = w * (uint32_t)T;
    if (W >= extent) return;
    // (everything this wave needs is requested at once: the first 64 entries of the run list too -- the list has T slots,
    // what lies behind its n_inner entries is never used -- so that one memory round trip stands between launch and stores)
    const uint16_t *rl = runs + (size_t)w * T;
    const uint32_t e0 = rl[lane];
    const uint32_t n_inner = winpart[w].n_inner;
    const uint32_t seam = (w == 0 || first_state[w] != last_state[w - 1]) ? 1u : 0u;
    // runs before this window: those of the earlier k_fin_windows blocks + the offset inside its block
    uint32_t before = 0;
    for (uint32_t b = lane; b < w / kFinBlock; b += 64u) before += fin[b].n_runs;
    const uint32_t idx0 = dpp_wave_sum_u32(before) + win_off[w];
    if (lane == 0 && seam) {
        if (idx0 < iv_cap) { iv[idx0].start = W; iv[idx0].state = first_state[w]; }
        if (idx0 > 0 && idx0 - 1 < iv_cap) iv[idx0 - 1].end = W;
    }
    for (uint32_t i = lane; i < n_inner; i += 64u) {
        const uint32_t e = i < 64u ? e0 : (uint32_t)rl[i];
        const uint32_t idx = idx0 + seam + i, start = W + (e & 0xFFFu);
        if (idx < iv_cap) { iv[idx].start = start; iv[idx].state = e >> 12; }
        if (idx > 0 && idx - 1 < iv_cap) iv[idx - 1].end = start;
    }
    if (w == n_win - 1 && lane == 0) {                     // the last run ends where classification ends
        const uint32_t last = idx0 + seam + n_inner;
        if (last > 0 && last - 1 < iv_cap) iv[last - 1].end = extent;
    }
}

// ---------------------------------------------------------------------------------------------
// config 5: site-list pileup (src/haplogroup/caller.rs:62-152): for every M/=/X base of a read with
// mapq >= min_quality whose 1-based position is a listed site, hist[site][4-bit base code] += 1.
//
// sorted_pos0 / sorted_idx: the sites sorted by 0-based position and their original indices; bucket[b]: index of
// the first sorted site with position >= 256*b.  One packed record per read (built on the host per call, like
// ReadRec): pos, CIGAR offset, low half of the base offset (the full offset = the block's 64-bit base + the 32-bit
// difference), mapq | n_cigar << 8 | n_bases << 16 with 255 / 0xFFFF meaning "the next record's offsets".
//
// A workgroup takes 256 consecutive reads.  A thread walks its read's first four CIGAR words (one 16-byte load) for
// the reference span and leaves at once when no site lies inside it (three reads in five at one site per ~300
// bases: no per-operation walk, no base is touched); hits go to a histogram of the workgroup's own sites in LDS
// (the reads are sorted, so they share a handful of sites) which is added to the global one once at the end --
// hits outside that range (unsorted input, a very long read) add to the global histogram directly.
// ---------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) SiteRec {
    int32_t  pos;
    uint32_t cigar_off;
    uint32_t seq_lo;
    uint32_t meta;
};
constexpr int kSiteLds = 64;            // sites a workgroup privatises

struct SiteArgs {
    const SiteRec *rec;                 // n + 1
    const unsigned long long *seq_base; // per workgroup of kBlock reads: base offset (in bases) of its first read
    const uint32_t *cigar;              // padded by 8 words
    const uint8_t  *seq4;
    uint32_t n;
    uint32_t min_quality, contig_len;
    unsigned long long ref_len;
    const uint32_t *sorted_pos0, *sorted_idx, *bucket;
    uint32_t n_buckets, n_sites;
    uint32_t *hist;
};

__global__ __launch_bounds__(kBlock) void k_site_pileup(SiteArgs a)
{
    __shared__ uint32_t s_hist[kSiteLds * 16];
    __shared__ uint32_t s_first;
    const uint32_t tid = threadIdx.x;
    const uint32_t r0 = blockIdx.x * kBlock;
    for (uint32_t i = tid; i < (uint32_t)kSiteLds * 16u; i += kBlock) s_hist[i] = 0;
    auto first_site_at = [&](unsigned long long x) {             // first sorted site with position >= x
        const unsigned long long bx = x >> 8;
        uint32_t lo = bx < a.n_buckets ? a.bucket[bx] : a.n_sites;
        while (lo < a.n_sites && a.sorted_pos0[lo] < x) ++lo;
        return lo;
    };
    if (tid == 0) {
        const int32_t p0 = a.rec[r0].pos;
        s_first = first_site_at(p0 < 0 ? 0ull : (unsigned long long)p0);
    }
    __syncthreads();
    const uint32_t first = s_first;
    const uint32_t r = r0 + tid;
    if (r < a.n) {
        const uint4 rr = *reinterpret_cast<const uint4 *>(a.rec + r);
        const uint32_t mq = rr.w & 255u;
        // fetch("chr:1-len"), caller.rs:33-36; the mapping-quality gate, caller.rs:80
        if ((uint32_t)rr.x < a.contig_len && mq >= a.min_quality) {
            uint32_t k = rr.y, k1 = k + ((rr.w >> 8) & 255u);
            unsigned long long slen = rr.w >> 16;
            if (((rr.w >> 8) & 255u) == 255u || slen == 0xFFFFull) {
                const uint4 nx = *reinterpret_cast<const uint4 *>(a.rec + r + 1);
                k1 = nx.y; slen = (uint32_t)(nx.z - rr.z);
            }
            const unsigned long long base = a.seq_base[blockIdx.x];
            const unsigned long long s0 = base + (uint32_t)(rr.z - (uint32_t)base);
            Q16 c4;
            __builtin_memcpy(&c4, a.cigar + k, 16);
            const uint32_t n = k1 - k;
            unsigned long long x = (uint32_t)rr.x, reflen = 0;
#pragma unroll
            for (uint32_t d = 0; d < 4u; ++d) {
                const uint32_t c = d < n ? c4.w[d] : 5u;
                reflen += ((0x18Du >> (c & 15u)) & 1u) ? (c >> 4) : 0u;
            }
            for (uint32_t kk = k + 4u; kk < k1; ++kk) {
                const uint32_t c = a.cigar[kk];
                reflen += ((0x18Du >> (c & 15u)) & 1u) ? (c >> 4) : 0u;
            }
            uint32_t lo = first_site_at(x);
            if (lo < a.n_sites && a.sorted_pos0[lo] < x + reflen) {      // some site inside the read's span: walk it
                unsigned long long y = 0;
                for (uint32_t kk = k; kk < k1; ++kk) {
                    const uint32_t d = kk - k;
                    const uint32_t c = d == 0 ? c4.w[0] : d == 1 ? c4.w[1] : d == 2 ? c4.w[2] : d == 3 ? c4.w[3] : a.cigar[kk];
                    const uint32_t op = c & 15u, l = c >> 4;
                    if (op_match(op)) {
                        while (lo < a.n_sites && a.sorted_pos0[lo] < x) ++lo;
                        for (; lo < a.n_sites && a.sorted_pos0[lo] < x + l; ++lo) {
                            const unsigned long long p = a.sorted_pos0[lo];
                            const unsigned long long qi = y + (p - x);
                            if (qi < slen && p < a.ref_len) {           // caller.rs:105,110-113
                                const unsigned long long bi = s0 + qi;
                                const uint32_t byte = a.seq4[bi >> 1];
                                const uint32_t code = (bi & 1ull) ? (byte & 15u) : (byte >> 4);
                                const uint32_t slot = lo - first;       // below `first`: wraps, goes to the global one
                                if (slot < (uint32_t)kSiteLds) atomicAdd(&s_hist[slot * 16u + code], 1u);
                                else atomicAdd(&a.hist[(unsigned long long)a.sorted_idx[lo] * 16ull + code], 1u);
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
    }
    __syncthreads();
    for (uint32_t i = tid; i < (uint32_t)kSiteLds * 16u; i += kBlock) {
        const uint32_t v = s_hist[i], si = first + (i >> 4);
        if (v && si < a.n_sites) atomicAdd(&a.hist[(unsigned long long)a.sorted_idx[si] * 16ull + (i & 15u)], v);
    }
}

} // namespace clk
