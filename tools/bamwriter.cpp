// Tooling only: a quick multi-threaded BAM writer for the end-to-end benchmark (one contig's decoded
// records -> coordinate-sorted BAM, no index).  Built on demand by tools/e2e_bench.py with g++.
#include <zlib.h>
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

static void put32(std::vector<uint8_t> &v, uint32_t x) { for (int i = 0; i < 4; ++i) v.push_back((uint8_t)(x >> (8 * i))); }
static void put16(std::vector<uint8_t> &v, uint16_t x) { v.push_back((uint8_t)x); v.push_back((uint8_t)(x >> 8)); }

static int reg2bin(int64_t beg, int64_t end)
{
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

static std::vector<uint8_t> bgzf_block(const uint8_t *d, size_t n, int level)
{
    std::vector<uint8_t> out(18 + compressBound(n) + 8);
    z_stream zs; memset(&zs, 0, sizeof(zs));
    deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
    zs.next_in = const_cast<uint8_t *>(d); zs.avail_in = (uInt)n;
    zs.next_out = out.data() + 18; zs.avail_out = (uInt)(out.size() - 18);
    deflate(&zs, Z_FINISH);
    const size_t clen = zs.total_out;
    deflateEnd(&zs);
    const uint8_t hdr[16] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0};
    memcpy(out.data(), hdr, 16);
    const uint16_t bsize = (uint16_t)(clen + 25);
    out[16] = (uint8_t)bsize; out[17] = (uint8_t)(bsize >> 8);
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), d, (uInt)n), isz = (uint32_t)n;
    memcpy(out.data() + 18 + clen, &crc, 4); memcpy(out.data() + 18 + clen + 4, &isz, 4);
    out.resize(18 + clen + 8);
    return out;
}

static void write_records(FILE *f, int32_t ref_id, uint64_t n,
                          const int32_t *pos, const uint16_t *flag, const uint8_t *mapq, const uint32_t *cigar_off,
                          const uint32_t *cigar, const uint64_t *qual_off, const uint8_t *qual, const uint32_t *qname_off,
                          const uint8_t *qname, int level, int threads);

extern "C" int tool_write_bam(const char *path, const char *header_text, const char *ref_name, uint32_t ref_len, uint64_t n,
                              const int32_t *pos, const uint16_t *flag, const uint8_t *mapq, const uint32_t *cigar_off,
                              const uint32_t *cigar, const uint64_t *qual_off, const uint8_t *qual, const uint32_t *qname_off,
                              const uint8_t *qname, int level, int threads)
{
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    std::vector<uint8_t> head;
    const std::string text = std::string(header_text) + "@SQ\tSN:" + ref_name + "\tLN:" + std::to_string(ref_len) + "\n";
    head.insert(head.end(), {'B', 'A', 'M', 1});
    put32(head, (uint32_t)text.size()); head.insert(head.end(), text.begin(), text.end());
    put32(head, 1); put32(head, (uint32_t)strlen(ref_name) + 1); head.insert(head.end(), ref_name, ref_name + strlen(ref_name) + 1); put32(head, ref_len);
    { auto b = bgzf_block(head.data(), head.size(), level); fwrite(b.data(), 1, b.size(), f); }
    write_records(f, 0, n, pos, flag, mapq, cigar_off, cigar, qual_off, qual, qname_off, qname, level, threads);
    { auto b = bgzf_block(nullptr, 0, level); fwrite(b.data(), 1, b.size(), f); }
    fclose(f);
    return 0;
}

// Multi-contig files are assembled by the caller: tool_write_bam_head (header block), then one
// tool_write_bam_body per reference in tid order (appends the reference's record blocks; the file
// size before the call << 16 is the reference's first virtual offset for the .bai), then tool_write_bam_eof.
extern "C" int tool_write_bam_head(const char *path, const char *header_text, int n_ref, const char *const *names, const uint32_t *lens, int level)
{
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    std::string text = header_text;
    for (int i = 0; i < n_ref; ++i) text += std::string("@SQ\tSN:") + names[i] + "\tLN:" + std::to_string(lens[i]) + "\n";
    std::vector<uint8_t> head;
    head.insert(head.end(), {'B', 'A', 'M', 1});
    put32(head, (uint32_t)text.size()); head.insert(head.end(), text.begin(), text.end());
    put32(head, (uint32_t)n_ref);
    for (int i = 0; i < n_ref; ++i) { put32(head, (uint32_t)strlen(names[i]) + 1); head.insert(head.end(), names[i], names[i] + strlen(names[i]) + 1); put32(head, lens[i]); }
    for (size_t a = 0; a < head.size(); a += 0xFF00) { auto b = bgzf_block(head.data() + a, std::min<size_t>(0xFF00, head.size() - a), level); fwrite(b.data(), 1, b.size(), f); }
    fclose(f);
    return 0;
}

extern "C" int tool_write_bam_body(const char *path, int32_t ref_id, uint64_t n,
                                   const int32_t *pos, const uint16_t *flag, const uint8_t *mapq, const uint32_t *cigar_off,
                                   const uint32_t *cigar, const uint64_t *qual_off, const uint8_t *qual, const uint32_t *qname_off,
                                   const uint8_t *qname, int level, int threads)
{
    FILE *f = fopen(path, "ab");
    if (!f) return -1;
    write_records(f, ref_id, n, pos, flag, mapq, cigar_off, cigar, qual_off, qual, qname_off, qname, level, threads);
    fclose(f);
    return 0;
}

extern "C" int tool_write_bam_eof(const char *path)
{
    FILE *f = fopen(path, "ab");
    if (!f) return -1;
    { auto b = bgzf_block(nullptr, 0, 1); fwrite(b.data(), 1, b.size(), f); }
    fclose(f);
    return 0;
}

static void write_records(FILE *f, int32_t ref_id, uint64_t n,
                          const int32_t *pos, const uint16_t *flag, const uint8_t *mapq, const uint32_t *cigar_off,
                          const uint32_t *cigar, const uint64_t *qual_off, const uint8_t *qual, const uint32_t *qname_off,
                          const uint8_t *qname, int level, int threads)
{
    // like htslib's bam_write1 (bgzf_flush_try): a record that does not fit into the current BGZF block
    // starts a new one, so no record straddles two blocks (set TOOL_BAM_STRADDLE=1 for 0xFF00-byte cuts)
    const bool straddle = getenv("TOOL_BAM_STRADDLE") && *getenv("TOOL_BAM_STRADDLE") == '1';
    const uint64_t kChunk = 200000;
    std::vector<uint8_t> carry;
    for (uint64_t a = 0; a < n; a += kChunk) {
        const uint64_t e = a + kChunk < n ? a + kChunk : n;
        std::vector<uint8_t> buf(std::move(carry));
        carry.clear();
        std::vector<size_t> cuts{0};                       // block boundaries inside buf
        size_t blk_begin = 0;
        for (uint64_t i = a; i < e; ++i) {
            const uint32_t nc = cigar_off[i + 1] - cigar_off[i], ln = qname_off[i + 1] - qname_off[i] + 1;
            const uint32_t ls = (uint32_t)(qual_off[i + 1] - qual_off[i]);
            const size_t rec_bytes = 36 + ln + 4ull * nc + (ls + 1) / 2 + ls;
            if (!straddle && buf.size() - blk_begin + rec_bytes > 0xFF00 && buf.size() > blk_begin) { cuts.push_back(buf.size()); blk_begin = buf.size(); }
            int64_t rlen = 0;
            for (uint32_t k = cigar_off[i]; k < cigar_off[i + 1]; ++k) { const uint32_t op = cigar[k] & 15; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += cigar[k] >> 4; }
            put32(buf, 32 + ln + 4 * nc + (ls + 1) / 2 + ls);
            put32(buf, (uint32_t)ref_id); put32(buf, (uint32_t)pos[i]);
            buf.push_back((uint8_t)ln); buf.push_back(mapq[i]); put16(buf, (uint16_t)reg2bin(pos[i], pos[i] + (rlen ? rlen : 1)));
            put16(buf, (uint16_t)nc); put16(buf, flag[i]); put32(buf, ls); put32(buf, 0xFFFFFFFFu); put32(buf, 0xFFFFFFFFu); put32(buf, 0);
            buf.insert(buf.end(), qname + qname_off[i], qname + qname_off[i + 1]); buf.push_back(0);
            const uint8_t *cp = (const uint8_t *)(cigar + cigar_off[i]);
            buf.insert(buf.end(), cp, cp + 4ull * nc);
            buf.insert(buf.end(), (ls + 1) / 2, (uint8_t)0x11);
            buf.insert(buf.end(), qual + qual_off[i], qual + qual_off[i + 1]);
        }
        if (straddle) {
            const size_t kB = 0xFF00, nbs = e == n ? (buf.size() + kB - 1) / kB : buf.size() / kB;
            cuts.clear();
            for (size_t b = 0; b <= nbs; ++b) cuts.push_back(std::min(b * kB, buf.size()));
            if (e != n) cuts.back() = nbs * kB;
        } else if (e == n) cuts.push_back(buf.size());   // the open block is closed at the end only
        // blocks [cuts[b], cuts[b+1]) ; what follows the last cut is carried into the next chunk
        const size_t nb = cuts.size() - 1;
        std::vector<std::vector<uint8_t>> outs(nb);
        std::atomic<size_t> next{0};
        auto work = [&]() { for (;;) { const size_t b = next.fetch_add(1); if (b >= nb) break; outs[b] = bgzf_block(buf.data() + cuts[b], cuts[b + 1] - cuts[b], level); } };
        std::vector<std::thread> th;
        for (int t = 1; t < threads; ++t) th.emplace_back(work);
        work();
        for (auto &t : th) t.join();
        for (auto &o : outs) fwrite(o.data(), 1, o.size(), f);
        if (cuts.back() < buf.size()) carry.assign(buf.begin() + (long)cuts.back(), buf.end());
    }
}
