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
constexpr unsigned long long kRowSparse = 1ull << 63;   // flag in a read's bit offset: query-order bits + its CIGAR (below)

// host views of the staged contig (callable_loci.hip fills these from the context's staging arrays).  cl_push_reads
// leaves every read with mapq >= min_mapq and a reference span as a string of pass bits in REFERENCE order -- bit j of
// the string <-> position pos + j: the pass bit of the M/=/X base there, 0 for D, N and bases without a quality byte --,
// word-aligned in `bits` from word off[i] to off[i + 1] (a plain match read: its thresholded quality string as it is; any
// other read: mapped through its CIGAR in the walk that validates the tile, where the operations are in the cache).  Only
// a read whose span dwarfs its query (long N gaps: span > 4 ql + 1024) stays in QUERY order, flagged kRowSparse, with
// its CIGAR in sc[sc_off[i], sc_off[i + 1]): it is walked per window.
struct RowReads {
    const int32_t *pos;
    const uint32_t *end;                          // pos + bam_cigar2rlen
    const uint8_t *mapq;
    const unsigned long long *off;                // n + 1 word offsets into `bits` (| kRowSparse)
    const uint64_t *bits;
    const uint32_t *sc_off, *sc;                  // CIGARs of the sparse reads (n + 1 offsets)
    uint32_t min_mapq;
};

struct RowCur { uint32_t k, k1, x, y, qlen, pos, end; unsigned long long o; bool sparse; };

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

// A read's pass bits from query order (qw: bit k <-> base k, ql bases, TWO readable words behind them) to reference
// order: rw[0, ceil(span / 64)) is written whole.  What the column walk visits as (alignment, qpos) with !is_del are the
// bases of M/=/X operations (mod.rs:30-37); a base without a quality byte (k >= ql) never passes.  Written without a
// branch on the kind of operation (insertions and deletions alternate at random in long reads: a mispredicted branch per
// operation cost more than the work): every operation appends `lr` bits to the reference string -- its length if it
// consumes the reference, else 0 -- of which the first `lq` come from the query bits -- if it is a match -- and the rest
// are zeros.  unmatched[0, *n_unmatched): where the inserted / clipped bases lie in the query (the caller's scratch holds
// nops entries), for unmatched_pass_sum.
struct QueryStretch { unsigned long long y, l; };
inline void ref_bits_from_query(const uint64_t *qw, unsigned long long ql, const uint32_t *cig, uint32_t nops, uint64_t *rw,
                                QueryStretch *unmatched, size_t *n_unmatched, unsigned long long *query_len)
{
    uint64_t acc = 0;
    uint32_t accn = 0;
    size_t wout = 0, nu = 0;
    unsigned long long y = 0;
    for (uint32_t j = 0; j < nops; ++j) {
        const uint32_t cw = cig[j], op = cw & 15u;
        const unsigned long long l = cw >> 4;
        const unsigned long long ism = (0x181u >> op) & 1u, radv = (0x18Du >> op) & 1u, qadv = (0x193u >> op) & 1u;
        unmatched[nu].y = y; unmatched[nu].l = l;                   // (kept only for I and S)
        nu += (size_t)(qadv & (ism ^ 1ull));
        const unsigned long long avail = y < ql ? ql - y : 0ull;
        unsigned long long have = std::min(avail, l) & (0ull - ism);   // bits that come from the query
        unsigned long long rem = l & (0ull - radv);                    // bits this operation appends
        unsigned long long src = std::min(y, ql);                      // (clamped: read, then masked away, when have = 0)
        y += l & (0ull - qadv);
        while (rem) {                                               // one trip unless the operation crosses a word
            const unsigned long long take = std::min<unsigned long long>(rem, 64u - accn);
            const unsigned long long tb = std::min(have, take);
            const unsigned long long w = src >> 6;
            const uint32_t s = (uint32_t)(src & 63ull);
            uint64_t v = (qw[w] >> s) | ((qw[w + 1] << 1) << (63u - s));
            v &= tb >= 64ull ? ~0ull : ((1ull << tb) - 1ull);
            acc |= v << accn;
            accn += (uint32_t)take; src += tb; have -= tb; rem -= take;
            if (accn == 64u) { rw[wout++] = acc; acc = 0; accn = 0; }
        }
    }
    if (accn) rw[wout++] = acc;
    *n_unmatched = nu;
    *query_len = y;
}

// The passing quality bytes of a read that are NOT bases of M/=/X operations: inserted and clipped bases (listed by
// ref_bits_from_query), and whatever the quality string holds beyond the CIGAR's query length (contig_profiler.rs:65-70
// sums over the matched bases only: the read's share of summed_baseq is the sum over its whole string minus this).
inline uint64_t unmatched_pass_sum(const uint8_t *q, unsigned long long ql, const QueryStretch *unmatched, size_t n_unmatched,
                                   unsigned long long query_len, uint8_t thr)
{
    uint64_t s = 0;
    if (ql == 0) return 0;
    for (size_t k = 0; k < n_unmatched; ++k) {
        const unsigned long long y = unmatched[k].y, l = unmatched[k].l;
        const unsigned long long cnt = y < ql ? std::min(ql - y, l) : 0ull;
        // the first four bases without a branch (insertions are short), the rest of a long one in a loop
        for (unsigned b = 0; b < 4u; ++b) {
            const uint8_t v = q[std::min(y + b, ql - 1ull)];
            s += (b < cnt && v >= thr) ? v : 0u;
        }
        for (unsigned long long b = 4; b < cnt; ++b) s += q[y + b] >= thr ? q[y + b] : 0u;
    }
    for (unsigned long long b = query_len; b < ql; ++b) s += q[b] >= thr ? q[b] : 0u;
    return s;
}

// a read that covers positions at or after W enters the sweep (reads below min_mapq never do, mod.rs:25; neither do
// reads that left no bits: no quality string)
inline void rows_enter(std::vector<RowCur> &act, const RowReads &H, uint32_t r, uint32_t W)
{
    if (H.mapq[r] < H.min_mapq) return;
    const unsigned long long o0 = H.off[r] & ~kRowSparse, o1 = H.off[r + 1] & ~kRowSparse;
    if (o1 == o0) return;
    RowCur cu;
    cu.pos = (uint32_t)H.pos[r]; cu.end = H.end[r];
    if (cu.end <= W || cu.end <= cu.pos) return;
    cu.o = o0 << 6;
    cu.sparse = (H.off[r] & kRowSparse) != 0ull;
    const unsigned long long nb = (o1 - o0) << 6;                   // bits stored (whole words: the tail is zeros)
    cu.qlen = nb > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)nb;
    cu.k = cu.k1 = 0; cu.x = cu.pos; cu.y = 0;
    if (cu.sparse) { cu.k = H.sc_off[r]; cu.k1 = H.sc_off[r + 1]; if (cu.k >= cu.k1) return; }
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
        if (!cu.sparse) {
            // -- reference-order bits: the window's stretch of the string, as it is
            const uint32_t sp = cu.pos > W ? cu.pos : W;
            unsigned long long tp = cu.end < Wend ? cu.end : Wend;
            tp = std::min<unsigned long long>(tp, (unsigned long long)cu.pos + cu.qlen);
            if (sp < tp) deposit_bits(H.bits, cu.o + (sp - cu.pos), (uint32_t)(tp - sp), out, r, sp - W);
            if (cu.end > Wend) act[keep++] = cu;
            continue;
        }
        // -- a sparse read: its M/=/X bases inside the window, operation by operation
        while (cu.k < cu.k1 && cu.x < Wend) {
            const uint32_t cw = H.sc[cu.k], op = cw & 15u, l = cw >> 4;
            const uint32_t radv = (0x18Du >> op) & 1u, qadv = (0x193u >> op) & 1u, ism = (0x181u >> op) & 1u;
            const uint32_t xe = cu.x + (radv ? l : 0u);
            if (xe < cu.x) { cu.k = cu.k1; break; }                 // wraps the 32-bit coordinate: flagged kErrRange at push
            if (ism) {
                const uint32_t sp = cu.x > W ? cu.x : W;
                const uint32_t lq = cu.y < cu.qlen ? std::min(cu.qlen - cu.y, l) : 0u;   // bases that have a (possibly zero) bit
                unsigned long long tp = xe < Wend ? xe : Wend;
                tp = std::min<unsigned long long>(tp, (unsigned long long)cu.x + lq);
                if (sp < tp) deposit_bits(H.bits, cu.o + cu.y + (sp - cu.x), (uint32_t)(tp - sp), out, r, sp - W);
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
