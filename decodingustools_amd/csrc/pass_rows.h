// pass_rows.h -- the pass-bit ROWS of the pileup kernel, built on the host at upload (plain C++, no HIP).
//
// qc_depth of a position (mod.rs:30-37) = the number of reads with mapq >= min_mapping_quality that have an M/=/X base
// there whose quality byte passes min_base_quality.  The byte test is taken once on the host (qual_pack.cpp: one bit
// per base, in query order, bit g of the contig's bit array <-> quality byte g); here the bits are laid out the way the
// kernel counts them: per window of T reference positions a stack of ROWS, a row = T bits, bit p of a row <-> reference
// position W + p.  Every read that covers the window gets one row in which no other read of the window overlaps it
// (reads arrive sorted by start: a row is free again when its last read has ended, and a new row is opened only when
// none is free, so a window has exactly as many rows as its deepest column of such reads); its M/=/X bases drop their
// pass bits at their reference positions, everything else (D, N, clipped and inserted bases, bases without a quality
// byte) stays 0.  The kernel then needs no CIGAR, no offsets and no shifts: qc_depth[p] = the number of rows with bit p
// set, a bit-sliced column sum over a coalesced stream.
//
// Memory layout of a window's rows: groups of 4 rows; a group = 64 blocks (32 positions each) x 4 rows x 32 bits =
// 1 KB, dword index (block << 2) | (row & 3) -- one 16-byte load per lane and group, 1 KB per wave instruction.
#pragma once
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

namespace dut {

constexpr uint32_t kRowGroupWords = 256;          // 32-bit words per group of 4 rows (T = 2048: 64 blocks x 4 rows)
constexpr uint32_t kRowLongOps = 64;              // reads with more CIGAR ops have (reference, query) checkpoints

// host views of the staged contig (callable_loci.hip fills these from the context's staging arrays)
struct RowReads {
    const int32_t *pos;
    const uint32_t *end;                          // pos + bam_cigar2rlen
    const uint8_t *mapq;
    const uint32_t *cigar_off, *cigar;
    const unsigned long long *qual_off;           // n + 1; also the bit offsets into `bits`
    const uint64_t *bits;                         // bit g = quality byte g passes min_base_quality
    const uint32_t *ck_x, *ck_y;                  // checkpoint before every 64th operation of the contig's CIGAR array
    uint32_t min_mapq;
};

struct RowCur { uint32_t k, k1, x, y, qlen, pos, end; unsigned long long q0; };

struct RowScratch { std::vector<uint32_t> rend; };

// n bits of `src` from bit offset o, OR-ed into row r of the window's groups at window-relative position d: up to 64
// bits a step (two destination words: the words of a row are 16 bytes apart, one per block of 32 positions)
inline void deposit_bits(const uint64_t *src, unsigned long long o, uint32_t n, uint32_t *grp, uint32_t r, uint32_t d)
{
    uint32_t *dst = grp + (size_t)(r >> 2) * kRowGroupWords + (r & 3u);
    while (n) {
        const uint32_t b = d >> 5, lo = d & 31u, take = std::min(64u - lo, n);
        const unsigned long long w = o >> 6;
        const uint32_t s = (uint32_t)(o & 63ull);
        uint64_t v = src[w] >> s;
        if (s + take > 64u) v |= src[w + 1] << (64u - s);
        if (take < 64u) v &= (1ull << take) - 1ull;
        v <<= lo;                                                   // lo + take <= 64
        dst[(size_t)b << 2] |= (uint32_t)v;
        if (lo + take > 32u) dst[((size_t)b + 1) << 2] |= (uint32_t)(v >> 32);
        o += take; d += take; n -= take;
    }
}

// a read that covers positions at or after W enters the sweep (reads below min_mapq never do, mod.rs:25); reads with
// more than kRowLongOps operations that start before W enter at their last checkpoint at or before W
inline void rows_enter(std::vector<RowCur> &act, const RowReads &H, uint32_t r, uint32_t W)
{
    if (H.mapq[r] < H.min_mapq) return;
    RowCur cu;
    cu.k = H.cigar_off[r]; cu.k1 = H.cigar_off[r + 1]; cu.x = (uint32_t)H.pos[r]; cu.y = 0;
    cu.pos = cu.x; cu.end = H.end[r];
    if (cu.k >= cu.k1 || cu.end <= W || cu.end <= cu.pos) return;
    const unsigned long long ql = H.qual_off[r + 1] - H.qual_off[r];
    cu.qlen = ql > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)ql;
    cu.q0 = H.qual_off[r];
    if (cu.k1 - cu.k > kRowLongOps && cu.x < W) {
        const uint32_t jlo = (cu.k + 63u) >> 6, jhi = (cu.k1 - 1u) >> 6;
        if (jlo <= jhi && H.ck_x[jlo] <= W) {
            uint32_t lo_j = jlo, hi_j = jhi;
            while (lo_j < hi_j) {
                const uint32_t mid = lo_j + ((hi_j - lo_j + 1u) >> 1);
                if (H.ck_x[mid] <= W) lo_j = mid; else hi_j = mid - 1u;
            }
            cu.k = lo_j << 6; cu.x = H.ck_x[lo_j]; cu.y = H.ck_y[lo_j];
        }
    }
    act.push_back(cu);
}

// One window [W, W + T): every cursor of `act` (in read order = start order) takes a row and drops the pass bits of its
// M/=/X bases inside the window into out[0, cap_groups * kRowGroupWords) (groups are zeroed as they are opened), then
// moves on; reads that have ended leave the list.  Returns the number of groups, or SIZE_MAX when cap_groups did not
// suffice (the list is then spoilt: the caller restores its copy).
template <uint32_t T>
inline size_t rows_window(std::vector<RowCur> &act, const RowReads &H, uint32_t W, uint32_t *out, size_t cap_groups, RowScratch &sc)
{
    static_assert(T == 64u * 32u, "a group holds 64 blocks of 32 positions");
    const unsigned long long Wend = (unsigned long long)W + T;
    std::vector<uint32_t> &rend = sc.rend;
    uint32_t nr = 0, rot = 0, min_end = 0xFFFFFFFFu;
    size_t ng = 0, keep = 0;
    const size_t na = act.size();
    for (size_t i = 0; i < na; ++i) {
        RowCur cu = act[i];
        if (cu.end <= W) continue;                                  // ended exactly at the seam: nothing here
        // -- the read's row: any row whose last read ended at or before this one's start (looked for from the row
        //    after the last one given out: with reads of one length the rows come free in the order they were taken)
        const uint32_t s = cu.pos > W ? cu.pos - W : 0u;
        const uint32_t e = cu.end < Wend ? cu.end - W : T;
        uint32_t r = nr;
        if (nr && min_end <= s) {                                   // (min_end: a lower bound of every row's end -- when even
            uint32_t mn = 0xFFFFFFFFu;                              //  that lies behind s no row is free: thousands of reads
            for (uint32_t j = 0; j < nr; ++j) {                     //  stacked on one position would else scan all rows each)
                uint32_t q = rot + j;
                if (q >= nr) q -= nr;
                if (rend[q] <= s) { r = q; break; }
                mn = std::min(mn, rend[q]);
            }
            if (r == nr) min_end = mn;                              // every row looked at: exact now
        }
        if (r == nr) {
            if ((nr & 3u) == 0u) {
                if (ng == cap_groups) return SIZE_MAX;
                memset(out + ng * kRowGroupWords, 0, kRowGroupWords * sizeof(uint32_t));
                ++ng;
            }
            if (rend.size() <= nr) rend.resize((size_t)nr + 64);
            ++nr;
        }
        rend[r] = e;                                                // (a row's end only grows: min_end stays a lower bound)
        min_end = std::min(min_end, e);
        rot = r + 1u == nr ? 0u : r + 1u;
        // -- its M/=/X bases inside the window (what the column walk visits as (alignment, qpos) with !is_del)
        while (cu.k < cu.k1 && cu.x < Wend) {
            const uint32_t cw = H.cigar[cu.k], op = cw & 15u, l = cw >> 4;
            const uint32_t radv = (0x18Du >> op) & 1u, qadv = (0x193u >> op) & 1u, ism = (0x181u >> op) & 1u;
            const uint32_t xe = cu.x + (radv ? l : 0u);
            if (xe < cu.x) { cu.k = cu.k1; break; }                 // wraps the 32-bit coordinate: flagged kErrRange at push
            if (ism) {
                const uint32_t sp = cu.x > W ? cu.x : W;
                const uint32_t lq = cu.y < cu.qlen ? std::min(cu.qlen - cu.y, l) : 0u;   // bases that have a quality byte
                unsigned long long tp = xe < Wend ? xe : Wend;
                tp = std::min<unsigned long long>(tp, (unsigned long long)cu.x + lq);
                if (sp < tp) deposit_bits(H.bits, cu.q0 + cu.y + (sp - cu.x), (uint32_t)(tp - sp), out, r, sp - W);
            }
            if (xe > Wend) break;                                   // the operation goes on in the next window
            cu.x = xe; cu.y += qadv ? l : 0u; cu.k += 1u;
        }
        if (cu.k < cu.k1 && cu.end > Wend) act[keep++] = cu;
    }
    act.resize(keep);
    return ng;
}

} // namespace dut
