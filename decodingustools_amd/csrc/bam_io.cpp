// bam_io.cpp -- implementation of include/dut_bam.h: BGZF/BAM/BAI and FASTA/FAI input and the
// file-level `coverage` driver.  Host-only code (zlib for the inflate); the per-position work is the
// device engine's (callable_loci.hip).
#include "../../include/dut_bam.h"
#include "../../include/dut_report.h"
#include "host_parallel.h"
#include "qual_pack.h"

#include <dlfcn.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

namespace {

void set_err(char *err, size_t n, const std::string &m)
{
    if (err && n) { snprintf(err, n, "%s", m.c_str()); }
}

inline uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline uint16_t rd16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }
inline uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }

// ---------------------------------------------------------------------------------------------
// BGZF: a series of gzip members with a BC extra field that gives the member size
// ---------------------------------------------------------------------------------------------
struct Bgzf {
    FILE *fp = nullptr;
    std::vector<uint8_t> cbuf, ubuf;      // compressed / uncompressed block
    uint64_t block_coff = 0;              // file offset of the current block
    uint64_t next_coff = 0;               // file offset of the next block
    size_t upos = 0;                      // read position inside ubuf
    bool eof = false;
    std::string err;

    bool load_block()
    {
        // returns false at EOF or on error (err set)
        for (;;) {
            block_coff = next_coff;
            if (fseeko(fp, (off_t)block_coff, SEEK_SET) != 0) { err = "seek failed"; return false; }
            uint8_t hdr[18];
            size_t got = fread(hdr, 1, 18, fp);
            if (got == 0) { eof = true; ubuf.clear(); upos = 0; return false; }
            if (got < 18 || hdr[0] != 31 || hdr[1] != 139 || hdr[2] != 8 || !(hdr[3] & 4)) { err = "not a BGZF block"; return false; }
            const uint16_t xlen = rd16(hdr + 10);
            // find the BC subfield (normally the only one: SI1='B', SI2='C', SLEN=2)
            std::vector<uint8_t> extra(xlen);
            memcpy(extra.data(), hdr + 12, std::min<size_t>(6, xlen));
            if (xlen > 6 && fread(extra.data() + 6, 1, xlen - 6, fp) != (size_t)(xlen - 6)) { err = "truncated BGZF header"; return false; }
            int bsize = -1;
            for (size_t i = 0; i + 4 <= extra.size();) {
                const uint16_t slen = rd16(&extra[i + 2]);
                if (extra[i] == 'B' && extra[i + 1] == 'C' && slen == 2 && i + 6 <= extra.size()) bsize = rd16(&extra[i + 4]);
                i += 4 + slen;
            }
            if (bsize < 0) { err = "BGZF block without BC field"; return false; }
            const size_t total = (size_t)bsize + 1;                  // whole member
            const size_t hlen = 12 + xlen;
            if (total < hlen + 8) { err = "bad BGZF block size"; return false; }
            const size_t clen = total - hlen - 8;
            cbuf.resize(clen + 8);
            if (fread(cbuf.data(), 1, clen + 8, fp) != clen + 8) { err = "truncated BGZF block"; return false; }
            const uint32_t isize = rd32(&cbuf[clen + 4]);
            if (isize > 65536) { err = "bad BGZF ISIZE"; return false; }
            ubuf.resize(isize);
            next_coff = block_coff + total;
            upos = 0;
            if (isize) {
                z_stream zs;
                memset(&zs, 0, sizeof(zs));
                if (inflateInit2(&zs, -15) != Z_OK) { err = "inflateInit2 failed"; return false; }
                zs.next_in = cbuf.data(); zs.avail_in = (uInt)clen;
                zs.next_out = ubuf.data(); zs.avail_out = isize;
                const int rc = inflate(&zs, Z_FINISH);
                inflateEnd(&zs);
                if (rc != Z_STREAM_END || zs.total_out != isize) { err = "inflate failed"; return false; }
                if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), ubuf.data(), isize) != rd32(&cbuf[clen])) { err = "BGZF CRC mismatch"; return false; }
                return true;
            }
            // empty block (e.g. the EOF marker): continue with the next one
        }
    }
    // read exactly n bytes; returns bytes read (< n only at EOF / error)
    size_t read(void *dst, size_t n)
    {
        uint8_t *d = (uint8_t *)dst;
        size_t done = 0;
        while (done < n) {
            if (upos >= ubuf.size()) { if (!load_block()) break; }
            const size_t take = std::min(n - done, ubuf.size() - upos);
            memcpy(d + done, ubuf.data() + upos, take);
            upos += take; done += take;
        }
        return done;
    }
    uint64_t tell() const { return upos >= ubuf.size() && !ubuf.empty() ? (next_coff << 16) : ((block_coff << 16) | (uint64_t)upos); }
    bool seek(uint64_t voff)
    {
        next_coff = voff >> 16; eof = false; err.clear();
        ubuf.clear(); upos = 0;
        if (!load_block()) return eof && (voff & 0xFFFF) == 0;
        upos = (size_t)(voff & 0xFFFF);
        return upos <= ubuf.size();
    }
};

struct RefSeq { std::string name; uint32_t len; };

inline double tnow() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline bool timing_on() { static const bool on = getenv("DUT_TIMING") && *getenv("DUT_TIMING") == '1'; return on; }

// (Giving back a chr21-sized contig's ~2.7 GB of decode buffers costs ~0.25 s of page-table work at close.
// Tried and dropped: transparent huge pages for these buffers -- with the usual `defrag = madvise` the
// first-touch faults compact synchronously and the parallel parse went from 0.14 to 0.73 s; dropping the
// pages with MADV_DONTNEED in slices on all threads before free() -- 0.34 s, the threads contend.)
// grow-only buffer without value initialisation (the decoded arrays are written exactly once)
template <class T>
struct RawBuf {
    T *p = nullptr;
    size_t n = 0, cap = 0;
    ~RawBuf() { free(p); }
    bool reserve(size_t need)
    {
        if (need <= cap) return true;
        size_t c = std::max<size_t>(need, cap + cap / 2 + 1024);
        T *q = (T *)realloc(p, c * sizeof(T));
        if (!q) return false;
        p = q; cap = c;
        return true;
    }
    void clear() { n = 0; }
};

using dut::parallel_for;

// libdeflate (what htslib itself prefers for BGZF) when the shared library is on the machine: about twice
// zlib's inflate speed and a carry-less-multiply CRC-32.  Bound at run time; zlib otherwise or with
// DUT_INFLATE=zlib.  Prototypes as published in libdeflate.h (1.x).
struct LibDeflate {
    void *(*alloc)() = nullptr;
    int (*decompress)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;
    void (*release)(void *) = nullptr;
    uint32_t (*crc)(uint32_t, const void *, size_t) = nullptr;
    bool ok = false;
};
const LibDeflate &libdeflate()
{
    static const LibDeflate ld = [] {
        LibDeflate l;
        const char *e = getenv("DUT_INFLATE");
        if (e && strcmp(e, "zlib") == 0) return l;
        void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) return l;
        l.alloc = (void *(*)())dlsym(h, "libdeflate_alloc_decompressor");
        l.decompress = (int (*)(void *, const void *, size_t, void *, size_t, size_t *))dlsym(h, "libdeflate_deflate_decompress");
        l.release = (void (*)(void *))dlsym(h, "libdeflate_free_decompressor");
        l.crc = (uint32_t (*)(uint32_t, const void *, size_t))dlsym(h, "libdeflate_crc32");
        l.ok = l.alloc && l.decompress && l.release && l.crc;
        return l;
    }();
    return ld;
}

// BGZF blocks inflated a batch at a time, the blocks of a batch in parallel (each block is an
// independent deflate stream), into one contiguous buffer the record parser walks.
struct BlockStream {
    FILE *fp = nullptr;
    uint64_t next_coff = 0;           // file offset of the next block to load
    size_t skip = 0;                  // bytes to skip in the first block after a seek
    bool valid = false, eof = false;
    std::string err;
    RawBuf<uint8_t> buf;              // inflated bytes; [cur, buf.n) not yet consumed
    size_t cur = 0;
    RawBuf<uint8_t> cbuf;
    // read-ahead: while a batch is inflated the next one is read (pread, own thread) into cbuf_next
    RawBuf<uint8_t> cbuf_next;
    dut::Thread ahead;
    uint64_t ahead_off = 0;
    size_t ahead_want = 0, ahead_got = 0;
    void drop_ahead() { if (ahead.joinable()) ahead.join(); ahead_want = 0; }
    ~BlockStream() { drop_ahead(); }
    // compressed bytes per fill: small right after a seek (a contig of a few reads must not cost a 32 MB
    // read + inflate), doubling up to 32 MB while the same stretch keeps being read
    static constexpr size_t kBatchMin = 256u << 10, kBatchMax = 32u << 20;
    size_t batch = kBatchMin;
    double t_read = 0, t_inflate = 0; // DUT_TIMING
    // offsets in buf at which the inflated blocks begin (ascending; htslib never lets a record straddle
    // two BGZF blocks, so in its files every block start is a record start and blocks can be walked
    // independently); `aligned` is cleared the first time a record is seen to cross a block start
    std::vector<size_t> bstart;
    bool aligned = true;

    void reset(uint64_t voff)
    {
        drop_ahead();
        next_coff = voff >> 16; skip = (size_t)(voff & 0xFFFF);
        buf.clear(); cur = 0; eof = false; err.clear(); valid = true;
        bstart.clear(); aligned = true;
        batch = kBatchMin;
    }
    // appends the next batch of blocks to buf (after dropping the consumed prefix); false at EOF or on error
    bool fill()
    {
        if (cur) {
            memmove(buf.p, buf.p + cur, buf.n - cur); buf.n -= cur;
            size_t k = 0;
            for (size_t v : bstart) if (v >= cur) bstart[k++] = v - cur;
            bstart.resize(k);
            cur = 0;
        }
        if (eof) return false;
        struct Blk { size_t in, clen, out; uint32_t isize, crc; };
        std::vector<Blk> blks;
        size_t got = 0, used = 0, out_total = 0;
        for (;;) {
            const double tr0 = tnow();
            if (ahead.joinable()) ahead.join();
            if (ahead_want && ahead_off == next_coff && ahead_want == batch) {
                std::swap(cbuf.p, cbuf_next.p); std::swap(cbuf.cap, cbuf_next.cap);
                got = ahead_got;
                ahead_want = 0;
            } else {
                ahead_want = 0;
                if (!cbuf.reserve(batch)) { err = "out of memory"; return false; }
                if (fseeko(fp, (off_t)next_coff, SEEK_SET) != 0) { err = "seek failed"; return false; }
                got = fread(cbuf.p, 1, batch, fp);
            }
            t_read += tnow() - tr0;
            if (got == 0) { eof = true; return false; }
            blks.clear(); used = 0; out_total = 0;
            while (used + 18 <= got) {
                const uint8_t *h = cbuf.p + used;
                if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) { err = "not a BGZF block"; return false; }
                const size_t xlen = rd16(h + 10);
                if (used + 12 + xlen > got) break;
                int bsize = -1;
                for (size_t i = 0; i + 4 <= xlen;) {
                    const uint8_t *x = h + 12 + i;
                    const size_t slen = rd16(x + 2);
                    if (x[0] == 'B' && x[1] == 'C' && slen == 2 && i + 6 <= xlen) bsize = rd16(x + 4);
                    i += 4 + slen;
                }
                if (bsize < 0) { err = "BGZF block without BC field"; return false; }
                const size_t total = (size_t)bsize + 1, hlen = 12 + xlen;
                if (total < hlen + 8) { err = "bad BGZF block size"; return false; }
                if (used + total > got) break;
                const uint32_t isize = rd32(h + total - 4);
                if (isize > 65536) { err = "bad BGZF ISIZE"; return false; }
                blks.push_back({used + hlen, total - hlen - 8, out_total, isize, rd32(h + total - 8)});
                out_total += isize;
                used += total;
            }
            if (!blks.empty()) break;
            if (got < batch) { err = "truncated BGZF block"; return false; }
            batch *= 2;                                   // a block larger than the batch cannot happen (<= 64 KiB); be safe
        }
        next_coff += used;
        if (batch < kBatchMax) batch *= 2;
        if (got >= used + 28 && cbuf_next.reserve(batch)) {            // more than the EOF marker follows: read on while this batch inflates
            ahead_off = next_coff; ahead_want = batch; ahead_got = 0;
            const int fd = fileno(fp);
            ahead = dut::spawn_or_run([this, fd]() {
                size_t n = 0;
                while (n < ahead_want) {
                    const ssize_t r = pread(fd, cbuf_next.p + n, ahead_want - n, (off_t)(ahead_off + n));
                    if (r <= 0) break;
                    n += (size_t)r;
                }
                ahead_got = n;
            });
        }
        const size_t base = buf.n;
        if (!buf.reserve(base + out_total)) { err = "out of memory"; return false; }
        buf.n = base + out_total;
        for (const Blk &bk : blks) if (bk.isize) bstart.push_back(base + bk.out);
        std::atomic<int> bad{0};
        const uint8_t *cp = cbuf.p;
        uint8_t *op = buf.p + base;
        // one z_stream per contiguous group of blocks
        const size_t grain = 16;
        const LibDeflate &ld = libdeflate();
        const double ti0 = tnow();
        parallel_for((blks.size() + grain - 1) / grain, 1, [&](size_t g) {
            if (ld.ok) {
                void *dec = ld.alloc();
                if (!dec) { bad = 1; return; }
                const size_t e = std::min(blks.size(), (g + 1) * grain);
                for (size_t i = g * grain; i < e; ++i) {
                    const Blk &b = blks[i];
                    if (!b.isize) continue;
                    size_t got_out = 0;
                    if (ld.decompress(dec, cp + b.in, b.clen, op + b.out, b.isize, &got_out) != 0 || got_out != b.isize) { bad = 2; break; }
                    if (ld.crc(0u, op + b.out, b.isize) != b.crc) { bad = 3; break; }
                }
                ld.release(dec);
                return;
            }
            z_stream zs;
            memset(&zs, 0, sizeof(zs));
            if (inflateInit2(&zs, -15) != Z_OK) { bad = 1; return; }
            const size_t e = std::min(blks.size(), (g + 1) * grain);
            for (size_t i = g * grain; i < e; ++i) {
                const Blk &b = blks[i];
                if (!b.isize) continue;
                inflateReset(&zs);
                zs.next_in = const_cast<uint8_t *>(cp + b.in); zs.avail_in = (uInt)b.clen;
                zs.next_out = op + b.out; zs.avail_out = b.isize;
                const int rc = inflate(&zs, Z_FINISH);
                if (rc != Z_STREAM_END || zs.total_out != b.isize) { bad = 2; break; }
                if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), op + b.out, b.isize) != b.crc) { bad = 3; break; }
            }
            inflateEnd(&zs);
        });
        t_inflate += tnow() - ti0;
        if (bad) { err = bad == 3 ? "BGZF CRC mismatch" : "inflate failed"; return false; }
        if (skip) {
            if (skip > buf.n - base) { err = "virtual offset beyond its block"; return false; }
            cur += skip; skip = 0;
        }
        return true;
    }
};

} // namespace

struct dut_bam {
    Bgzf z;
    std::string path, err, text;
    std::vector<RefSeq> refs;
    uint64_t data_start = 0;                 // virtual offset of the first record
    // .bai: smallest chunk start per reference
    bool has_index = false;
    std::vector<uint64_t> ref_start;         // UINT64_MAX = no records
    std::vector<int64_t> ref_mapped;         // mapped reads per reference from the index metadata, -1 = not recorded
    // sequential state
    bool pending = false;                    // `rec` holds a record that was read but not consumed
    std::vector<uint8_t> rec;
    int32_t last_tid_done = -1;
    BlockStream st;                          // the record stream of dut_bam_read_contig
    // SoA of the last contig read
    RawBuf<int32_t> pos;
    RawBuf<uint16_t> flag;
    RawBuf<uint8_t> mapq, qual, qname, seq4;
    RawBuf<uint32_t> cigar_off, cigar, qname_off;
    RawBuf<uint64_t> qual_off, seq_off;
    // dut_bam_read_contig_bits: the base-quality test taken at parse (one bit per base) and the reads' sums
    RawBuf<uint64_t> qbits;
    RawBuf<uint32_t> qsum;
};

namespace {

bool read_record(dut_bam *b)
{
    uint8_t szb[4];
    const size_t g = b->z.read(szb, 4);
    if (g == 0) return false;                               // clean EOF
    if (g < 4) { b->err = b->z.err.empty() ? "truncated BAM record" : b->z.err; return false; }
    const uint32_t bs = rd32(szb);
    if (bs < 32 || bs > (1u << 29)) { b->err = "bad BAM block_size"; return false; }
    b->rec.resize(bs);
    if (b->z.read(b->rec.data(), bs) != bs) { b->err = b->z.err.empty() ? "truncated BAM record" : b->z.err; return false; }
    return true;
}

bool load_bai(dut_bam *b)
{
    std::string p1 = b->path + ".bai", p2 = b->path;
    if (p2.size() > 4 && p2.substr(p2.size() - 4) == ".bam") p2 = p2.substr(0, p2.size() - 4) + ".bai";
    FILE *f = fopen(p1.c_str(), "rb");
    if (!f) f = fopen(p2.c_str(), "rb");
    if (!f) return false;
    std::vector<uint8_t> d;
    uint8_t buf[65536];
    size_t g;
    while ((g = fread(buf, 1, sizeof(buf), f)) > 0) d.insert(d.end(), buf, buf + g);
    fclose(f);
    if (d.size() < 8 || memcmp(d.data(), "BAI\1", 4) != 0) return false;
    size_t o = 4;
    const uint32_t n_ref = rd32(&d[o]); o += 4;
    std::vector<uint64_t> start(n_ref, UINT64_MAX);
    std::vector<int64_t> mapped(n_ref, -1);
    for (uint32_t r = 0; r < n_ref; ++r) {
        if (o + 4 > d.size()) return false;
        const uint32_t n_bin = rd32(&d[o]); o += 4;
        for (uint32_t i = 0; i < n_bin; ++i) {
            if (o + 8 > d.size()) return false;
            const uint32_t bin = rd32(&d[o]), n_chunk = rd32(&d[o + 4]); o += 8;
            if (o + 16ull * n_chunk > d.size()) return false;
            if (bin != 37450)                                 // the metadata pseudo-bin
                for (uint32_t c = 0; c < n_chunk; ++c) start[r] = std::min(start[r], rd64(&d[o + 16ull * c]));
            else if (n_chunk >= 2) mapped[r] = (int64_t)rd64(&d[o + 16]);       // second pseudo-chunk: n_mapped, n_unmapped
            o += 16ull * n_chunk;
        }
        if (o + 4 > d.size()) return false;
        const uint32_t n_intv = rd32(&d[o]); o += 4;
        if (o + 8ull * n_intv > d.size()) return false;
        o += 8ull * n_intv;
    }
    if (n_ref != b->refs.size()) return false;
    b->ref_start.swap(start);
    b->ref_mapped.swap(mapped);
    return true;
}

// .csi (what `samtools index -c` writes, needed for references longer than 2^29): BGZF-compressed; magic, min_shift,
// depth, l_aux + aux, n_ref, then per reference the bins with {bin, loffset, n_chunk, chunks}.  Only what the
// .bai gives is taken: the smallest chunk start per reference and the mapped-read count of the metadata
// pseudo-bin, whose number depends on the depth: ((1 << 3 * (depth + 1)) - 1) / 7 + 1.
bool load_csi(dut_bam *b)
{
    std::string p1 = b->path + ".csi", p2 = b->path;
    if (p2.size() > 4 && p2.substr(p2.size() - 4) == ".bam") p2 = p2.substr(0, p2.size() - 4) + ".csi";
    Bgzf z;
    z.fp = fopen(p1.c_str(), "rb");
    if (!z.fp) z.fp = fopen(p2.c_str(), "rb");
    if (!z.fp) return false;
    std::vector<uint8_t> d;
    uint8_t buf[65536];
    size_t g;
    while ((g = z.read(buf, sizeof(buf))) > 0) { d.insert(d.end(), buf, buf + g); if (d.size() > (1ull << 31)) break; }
    fclose(z.fp); z.fp = nullptr;
    if (!z.err.empty() || d.size() < 20 || memcmp(d.data(), "CSI\1", 4) != 0) return false;
    const int32_t depth = (int32_t)rd32(&d[8]), l_aux = (int32_t)rd32(&d[12]);
    if (depth < 0 || depth > 9 || l_aux < 0) return false;
    size_t o = 16 + (size_t)l_aux;
    if (o + 4 > d.size()) return false;
    const uint32_t n_ref = rd32(&d[o]); o += 4;
    const uint32_t meta_bin = (uint32_t)((((uint64_t)1 << (3 * (depth + 1))) - 1) / 7 + 1);
    std::vector<uint64_t> start(n_ref, UINT64_MAX);
    std::vector<int64_t> mapped(n_ref, -1);
    for (uint32_t r = 0; r < n_ref; ++r) {
        if (o + 4 > d.size()) return false;
        const uint32_t n_bin = rd32(&d[o]); o += 4;
        for (uint32_t i = 0; i < n_bin; ++i) {
            if (o + 16 > d.size()) return false;
            const uint32_t bin = rd32(&d[o]), n_chunk = rd32(&d[o + 12]); o += 16;      // bin, loffset (8), n_chunk
            if (o + 16ull * n_chunk > d.size()) return false;
            if (bin != meta_bin)
                for (uint32_t c = 0; c < n_chunk; ++c) start[r] = std::min(start[r], rd64(&d[o + 16ull * c]));
            else if (n_chunk >= 2) mapped[r] = (int64_t)rd64(&d[o + 16]);
            o += 16ull * n_chunk;
        }
    }
    if (n_ref != b->refs.size()) return false;
    b->ref_start.swap(start);
    b->ref_mapped.swap(mapped);
    return true;
}

} // namespace

extern "C" {

dut_bam *dut_bam_open(const char *path, char *err, size_t err_len)
{
    if (!path) { set_err(err, err_len, "null path"); return nullptr; }
    dut_bam *b = new dut_bam();
    b->path = path;
    b->z.fp = fopen(path, "rb");
    if (!b->z.fp) { set_err(err, err_len, std::string("cannot open ") + path); delete b; return nullptr; }
    uint8_t m[12];
    auto bad = [&](const std::string &why) { set_err(err, err_len, why + (b->z.err.empty() ? "" : ": " + b->z.err)); dut_bam_close(b); return (dut_bam *)nullptr; };
    if (b->z.read(m, 8) != 8 || memcmp(m, "BAM\1", 4) != 0) return bad("not a BAM file");
    const uint32_t l_text = rd32(m + 4);
    b->text.resize(l_text);
    if (l_text && b->z.read(&b->text[0], l_text) != l_text) return bad("truncated BAM header");
    if (b->z.read(m, 4) != 4) return bad("truncated BAM header");
    const uint32_t n_ref = rd32(m);
    for (uint32_t i = 0; i < n_ref; ++i) {
        if (b->z.read(m, 4) != 4) return bad("truncated BAM header");
        const uint32_t l_name = rd32(m);
        std::string nm(l_name, '\0');
        if (l_name == 0 || b->z.read(&nm[0], l_name) != l_name || b->z.read(m, 4) != 4) return bad("truncated BAM header");
        nm.resize(strlen(nm.c_str()));
        b->refs.push_back({nm, rd32(m)});
    }
    b->data_start = b->z.tell();
    b->has_index = load_bai(b) || load_csi(b);
    return b;
}

void dut_bam_close(dut_bam *b)
{
    if (!b) return;
    b->st.drop_ahead();                      // a read-ahead may still be using the descriptor
    if (b->z.fp) fclose(b->z.fp);
    delete b;
}

const char *dut_bam_error(const dut_bam *b) { return b ? b->err.c_str() : "null reader"; }
int dut_bam_n_ref(const dut_bam *b) { return b ? (int)b->refs.size() : 0; }
const char *dut_bam_ref_name(const dut_bam *b, int tid) { return (b && tid >= 0 && (size_t)tid < b->refs.size()) ? b->refs[tid].name.c_str() : nullptr; }
uint32_t dut_bam_ref_len(const dut_bam *b, int tid) { return (b && tid >= 0 && (size_t)tid < b->refs.size()) ? b->refs[tid].len : 0; }
const char *dut_bam_header_text(const dut_bam *b, size_t *len) { if (len) *len = b ? b->text.size() : 0; return b ? b->text.data() : nullptr; }
int dut_bam_has_index(const dut_bam *b) { return b && b->has_index ? 1 : 0; }
int64_t dut_bam_ref_mapped(const dut_bam *b, int tid)
{
    return (b && b->has_index && tid >= 0 && (size_t)tid < b->ref_mapped.size()) ? b->ref_mapped[tid] : -1;
}

// bits_thr >= 0: the packed variant -- instead of copying a record's quality bytes the parser tests them against
// min_base_quality = bits_thr (mod.rs:33) while they are in the cache, leaves one bit per base (bit d_qual + k of qbits)
// and the sum of the passing values over the bases of the record's M/=/X operations (contig_profiler.rs:65-70)
static int dut_bam_read_contig_impl(dut_bam *b, int tid, dut_records *out, const uint64_t **seq_off, const uint8_t **seq4, int bits_thr = -1)
{
    if (!b || !out || tid < 0 || (size_t)tid >= b->refs.size()) return CL_ERR_INVALID;
    const bool want_bits = bits_thr >= 0;
    const uint8_t thr = (uint8_t)(want_bits ? bits_thr : 0);
    const int plevel = dut::qual_pack_level();
    b->qbits.clear(); b->qsum.clear();
    b->err.clear();
    b->pos.clear(); b->flag.clear(); b->mapq.clear(); b->qual.clear(); b->qname.clear(); b->cigar.clear(); b->seq4.clear();
    b->cigar_off.clear(); b->qname_off.clear(); b->qual_off.clear(); b->seq_off.clear();
    const bool want_seq = seq_off && seq4;
    BlockStream &st = b->st;
    st.fp = b->z.fp;
    bool nothing = false;
    // position the stream at the first record that can belong to tid
    if (b->has_index) {
        if (b->ref_start[tid] == UINT64_MAX) nothing = true;            // no records at all: read nothing
        else st.reset(b->ref_start[tid]);
    } else if (!st.valid || tid <= b->last_tid_done) st.reset(b->data_start);   // forward-only: rewind for an earlier contig

    uint64_t n = 0, n_cig = 0, n_qual = 0, n_name = 0, n_bases = 0;
    double t_scan = 0, t_parse = 0;
    st.t_read = st.t_inflate = 0;
    struct RecInfo { size_t off; uint32_t n_cigar, l_seq, l_name; size_t cig_off; uint64_t d_cig, d_qual, d_name, d_base; };
    std::vector<RecInfo> recs;
    // the fields of one record the arrays need (r = first byte after block_size); false = malformed
    auto decode = [](const uint8_t *r, uint32_t bs, RecInfo &x) -> bool {
        const uint32_t l_read_name = r[8];
        uint32_t n_cigar = rd16(r + 12);
        const uint32_t l_seq = rd32(r + 16);
        size_t q = 32;
        if (q + l_read_name + 4ull * n_cigar + (l_seq + 1) / 2 + (uint64_t)l_seq > bs || l_read_name == 0) return false;
        size_t cig_off = q + l_read_name;
        const uint8_t *cig = r + cig_off;
        q = cig_off + 4ull * n_cigar + (l_seq + 1) / 2 + l_seq;
        // long CIGARs live in the CG:B,I tag behind a <l_seq>S<reflen>N placeholder
        if (n_cigar == 2 && (rd32(cig) & 15u) == 4 && (rd32(cig) >> 4) == l_seq && (rd32(cig + 4) & 15u) == 3) {
            size_t a = q;
            while (a + 3 <= bs) {
                const uint8_t t0 = r[a], t1 = r[a + 1], ty = r[a + 2];
                a += 3;
                size_t len = 0;
                auto elt = [](uint8_t c) -> size_t { switch (c) { case 'c': case 'C': case 'A': return 1; case 's': case 'S': return 2; case 'i': case 'I': case 'f': return 4; default: return 0; } };
                if (ty == 'Z' || ty == 'H') { while (a + len < bs && r[a + len]) ++len; len += 1; }
                else if (ty == 'B') {
                    if (a + 5 > bs) break;
                    const uint8_t sub = r[a]; const uint32_t cnt = rd32(r + a + 1);
                    if (t0 == 'C' && t1 == 'G' && sub == 'I' && a + 5 + 4ull * cnt <= bs) { cig_off = a + 5; n_cigar = cnt; break; }
                    len = 5 + elt(sub) * (size_t)cnt;
                } else { len = elt(ty); if (!len) break; }
                a += len;
            }
        }
        x.n_cigar = n_cigar; x.l_seq = l_seq; x.l_name = l_read_name - 1; x.cig_off = cig_off;
        return true;
    };
    int32_t *d_pos = nullptr; uint16_t *d_flag = nullptr; uint8_t *d_mapq = nullptr;
    uint32_t *d_coff = nullptr, *d_cig = nullptr, *d_noff = nullptr;
    uint64_t *d_qoff = nullptr, *d_soff = nullptr;
    uint8_t *d_qual = nullptr, *d_name = nullptr, *d_seq = nullptr;
    uint64_t *d_qbits = nullptr; uint32_t *d_qsum = nullptr;
    // room for the totals so far; the destination pointers are refreshed (the buffers may move)
    auto make_room = [&]() -> bool {
        const size_t seq_bytes_old = b->seq4.n, seq_bytes_new = want_seq ? (size_t)((n_bases + 1) / 2) : 0;
        if (!b->pos.reserve(n) || !b->flag.reserve(n) || !b->mapq.reserve(n) || !b->cigar_off.reserve(n + 1) || !b->qual_off.reserve(n + 1) ||
            !b->qname_off.reserve(n + 1) || !b->seq_off.reserve(n + 1) || !b->cigar.reserve(n_cig) || !b->qual.reserve(want_bits ? 1 : n_qual) ||
            !b->qname.reserve(n_name) || !b->seq4.reserve(seq_bytes_new + 1)) { b->err = "out of memory"; return false; }
        if (want_seq && seq_bytes_new > seq_bytes_old) memset(b->seq4.p + seq_bytes_old, 0, seq_bytes_new - seq_bytes_old);
        b->seq4.n = seq_bytes_new;
        if (want_bits) {
            // the bit array grows zeroed (records OR their bits in, the words at record seams from two threads)
            const size_t words_old = b->qbits.n, words_new = (size_t)((n_qual + 63) >> 6) + 2;
            if (!b->qbits.reserve(words_new) || !b->qsum.reserve(n + 1)) { b->err = "out of memory"; return false; }
            if (words_new > words_old) memset(b->qbits.p + words_old, 0, (words_new - words_old) * sizeof(uint64_t));
            b->qbits.n = words_new;
            d_qbits = b->qbits.p; d_qsum = b->qsum.p;
        }
        d_pos = b->pos.p; d_flag = b->flag.p; d_mapq = b->mapq.p;
        d_coff = b->cigar_off.p; d_cig = b->cigar.p; d_noff = b->qname_off.p;
        d_qoff = b->qual_off.p; d_soff = b->seq_off.p;
        d_qual = b->qual.p; d_name = b->qname.p; d_seq = b->seq4.p;
        return true;
    };
    // one record into the arrays (idx = its ordinal in the contig, x.d_* = its offsets)
    auto store = [&](const uint8_t *r, const RecInfo &x, uint64_t idx) {
        d_pos[idx] = (int32_t)rd32(r + 4);
        d_flag[idx] = rd16(r + 14);
        d_mapq[idx] = r[9];
        d_coff[idx] = (uint32_t)x.d_cig; d_qoff[idx] = x.d_qual; d_noff[idx] = (uint32_t)x.d_name; d_soff[idx] = x.d_base;
        memcpy(d_cig + x.d_cig, r + x.cig_off, 4ull * x.n_cigar);
        memcpy(d_name + x.d_name, r + 32, x.l_name);
        const uint8_t *pk = r + 32 + x.l_name + 1 + 4ull * rd16(r + 12);
        if (!want_bits) memcpy(d_qual + x.d_qual, pk + (x.l_seq + 1) / 2, x.l_seq);
        else {
            // the quality bytes, tested where they lie: pass words (zeros above l_seq) and the sum over the whole string ...
            static thread_local std::vector<uint64_t> lw;
            const uint8_t *q = pk + (x.l_seq + 1) / 2;
            const size_t nw = ((size_t)x.l_seq + 63) >> 6;
            if (lw.size() < nw + 1) lw.resize(nw + 1);
            uint64_t total = x.l_seq ? dut::qual_pass_read(q, x.l_seq, thr, lw.data(), plevel) : 0;
            // ... minus the passing values of inserted / clipped bases and of whatever lies beyond the CIGAR's query length
            if (total) {
                const uint8_t *cg = r + x.cig_off;
                uint64_t y = 0, minus = 0;
                for (uint32_t j = 0; j < x.n_cigar; ++j) {
                    const uint32_t cw = rd32(cg + 4ull * j), op = cw & 15u, l = cw >> 4;
                    if (!((0x193u >> op) & 1u)) continue;
                    if (!((0x181u >> op) & 1u))
                        for (uint64_t k = y, e = std::min<uint64_t>(x.l_seq, y + l); k < e; ++k) minus += q[k] >= thr ? q[k] : 0u;
                    y += l;
                }
                for (uint64_t k = y; k < x.l_seq; ++k) minus += q[k] >= thr ? q[k] : 0u;
                total -= minus;
            }
            d_qsum[idx] = (uint32_t)std::min<uint64_t>(total, 0xFFFFFFFFull);
            const uint64_t B = x.d_qual;
            const uint32_t sh = (uint32_t)(B & 63ull);
            uint64_t *dst = d_qbits + (B >> 6);
            for (size_t k = 0; k < nw; ++k) {
                const uint64_t v = lw[k];
                if (!v) continue;
                __atomic_fetch_or(&dst[k], v << sh, __ATOMIC_RELAXED);
                if (sh && (v >> (64u - sh))) __atomic_fetch_or(&dst[k + 1], v >> (64u - sh), __ATOMIC_RELAXED);
            }
        }
        if (want_seq && x.l_seq) {
            // nibble-continuous store: bytes shared with the neighbouring records are OR-ed atomically
            auto code = [&](uint32_t i) -> uint8_t { return (i & 1u) ? (uint8_t)(pk[i >> 1] & 15u) : (uint8_t)(pk[i >> 1] >> 4); };
            const uint64_t B = x.d_base;
            const uint32_t l = x.l_seq;
            uint32_t i = 0;
            if (B & 1u) { __atomic_fetch_or(&d_seq[B >> 1], code(0), __ATOMIC_RELAXED); i = 1; }
            const uint64_t byte0 = (B + i) >> 1;
            const uint32_t full = (l - i) >> 1;
            if (i == 0) memcpy(d_seq + byte0, pk, full);
            else for (uint32_t k2 = 0; k2 < full; ++k2) d_seq[byte0 + k2] = (uint8_t)((pk[k2] << 4) | (pk[k2 + 1] >> 4));
            i += 2 * full;
            if (i < l) __atomic_fetch_or(&d_seq[(B + i) >> 1], (uint8_t)(code(i) << 4), __ATOMIC_RELAXED);
        }
    };
    struct Unit { size_t s, e, cut; uint64_t n, n_cig, n_qual, n_name; int status; };   // status: 0 ok, 1 crosses the block end, 2 malformed
    std::vector<Unit> units;
    bool end = nothing;
    bool need_fill = st.cur >= st.buf.n;                                  // else: records left over from the previous contig's window
    while (!end) {
        if (need_fill && !st.fill()) {
            if (!st.err.empty()) { b->err = st.err; return CL_ERR_INVALID; }
            if (st.cur < st.buf.n) { b->err = "truncated BAM record"; return CL_ERR_INVALID; }
            break;                                                        // clean EOF
        }
        need_fill = true;
        const uint8_t *buf = st.buf.p;
        const size_t size = st.buf.n;
        const uint64_t n0 = n;
        bool done_parallel = false;
        if (st.aligned && size > st.cur) {
            // ---- blocks walked independently: count, place, store (all three in parallel over the blocks) ----
            const double ts0 = tnow();
            units.clear();
            size_t s0 = st.cur;
            for (size_t v : st.bstart) if (v > s0) { units.push_back({s0, v, SIZE_MAX, 0, 0, 0, 0, 0}); s0 = v; }
            units.push_back({s0, size, SIZE_MAX, 0, 0, 0, 0, 0});
            parallel_for(units.size(), 8, [&](size_t ui) {
                Unit &u = units[ui];
                size_t o = u.s;
                while (o < u.e) {
                    if (o + 4 > u.e) { u.status = 1; return; }
                    const uint32_t bs = rd32(buf + o);
                    if (bs < 32 || bs > (1u << 29)) { u.status = 2; return; }
                    if (o + 4 + (size_t)bs > u.e) { u.status = 1; return; }
                    const uint8_t *r = buf + o + 4;
                    const int32_t ref_id = (int32_t)rd32(r);
                    if (ref_id < 0 || ref_id > tid) { u.cut = o; return; }
                    if (ref_id == tid) {
                        RecInfo x;
                        if (!decode(r, bs, x)) { u.status = 2; return; }
                        u.n += 1; u.n_cig += x.n_cigar; u.n_qual += x.l_seq; u.n_name += x.l_name;
                    }
                    o += 4 + (size_t)bs;
                }
            });
            // the units up to the first one that meets the next contig; a unit that is not a whole number of
            // records before that point means this file lets records straddle blocks: the general walk takes over
            size_t n_units = units.size();
            bool straddle = false, malformed = false;
            for (size_t ui = 0; ui < units.size(); ++ui) {
                if (units[ui].status == 1) { straddle = true; break; }
                if (units[ui].status == 2) { malformed = true; break; }
                if (units[ui].cut != SIZE_MAX) { n_units = ui + 1; break; }
            }
            if (straddle || malformed) st.aligned = false;        // (a malformed record is reported by the general walk)
            else {
                std::vector<uint64_t> ub(4 * n_units);
                for (size_t ui = 0; ui < n_units; ++ui) {
                    ub[4 * ui] = n; ub[4 * ui + 1] = n_cig; ub[4 * ui + 2] = n_qual; ub[4 * ui + 3] = n_name;
                    n += units[ui].n; n_cig += units[ui].n_cig; n_qual += units[ui].n_qual; n_name += units[ui].n_name;
                }
                n_bases = n_qual;
                if (n_cig > 0xFFFFFFF0ull || n_name > 0xFFFFFFF0ull) { b->err = "contig too large for 32-bit offsets"; return CL_ERR_RANGE; }
                t_scan += tnow() - ts0;
                const double tp0 = tnow();
                if (!make_room()) return CL_ERR_INVALID;
                parallel_for(n_units, 8, [&](size_t ui) {
                    const Unit &u = units[ui];
                    const size_t stop = u.cut != SIZE_MAX ? u.cut : u.e;
                    uint64_t idx = ub[4 * ui], c = ub[4 * ui + 1], q = ub[4 * ui + 2], nm = ub[4 * ui + 3];
                    for (size_t o = u.s; o < stop;) {
                        const uint32_t bs = rd32(buf + o);
                        const uint8_t *r = buf + o + 4;
                        if ((int32_t)rd32(r) == tid) {
                            RecInfo x;
                            decode(r, bs, x);
                            x.d_cig = c; x.d_qual = q; x.d_name = nm; x.d_base = q;
                            store(r, x, idx);
                            idx += 1; c += x.n_cigar; q += x.l_seq; nm += x.l_name;
                        }
                        o += 4 + (size_t)bs;
                    }
                });
                const Unit &last = units[n_units - 1];
                if (last.cut != SIZE_MAX) { st.cur = last.cut; end = true; }
                else st.cur = last.e;
                b->pos.n = b->flag.n = b->mapq.n = n;
                b->cigar.n = n_cig; b->qual.n = want_bits ? 0 : n_qual; b->qname.n = n_name; b->qsum.n = want_bits ? n : 0;
                t_parse += tnow() - tp0;
                done_parallel = true;
            }
        }
        if (done_parallel) continue;
        // ---- general walk: pass 1, record boundaries of this window and destination offsets (sequential) ----
        const double ts0 = tnow();
        recs.clear();
        size_t o = st.cur;
        while (o + 4 <= size) {
            const uint32_t bs = rd32(buf + o);
            if (bs < 32 || bs > (1u << 29)) { b->err = "bad BAM block_size"; return CL_ERR_INVALID; }
            if (o + 4 + (size_t)bs > size) break;                         // needs the next batch
            // the chain of block_size fields is a dependent walk with one cache miss per record: most
            // records have their predecessor's size, so the headers a few records ahead are prefetched
            __builtin_prefetch(buf + o + 6 * (4 + (size_t)bs));
            __builtin_prefetch(buf + o + 12 * (4 + (size_t)bs));
            const uint8_t *r = buf + o + 4;
            const int32_t ref_id = (int32_t)rd32(r);
            if (ref_id < 0 || ref_id > tid) { end = true; break; }        // sorted: past this contig (the record stays in the stream)
            if (ref_id == tid) {
                RecInfo x;
                if (!decode(r, bs, x)) { b->err = "malformed BAM record"; return CL_ERR_INVALID; }
                x.off = o + 4; x.d_cig = n_cig; x.d_qual = n_qual; x.d_name = n_name; x.d_base = n_bases;
                recs.push_back(x);
                n += 1; n_cig += x.n_cigar; n_qual += x.l_seq; n_name += x.l_name; n_bases += x.l_seq;
                if (n_cig > 0xFFFFFFF0ull || n_name > 0xFFFFFFF0ull) { b->err = "contig too large for 32-bit offsets"; return CL_ERR_RANGE; }
            }
            o += 4 + (size_t)bs;
        }
        t_scan += tnow() - ts0;
        const double tp0 = tnow();
        // ---- pass 2: fill the arrays, records in parallel ----
        if (!make_room()) return CL_ERR_INVALID;
        const RecInfo *ri = recs.data();
        parallel_for(recs.size(), 8192, [&](size_t k) { store(buf + ri[k].off, ri[k], n0 + k); });
        b->pos.n = b->flag.n = b->mapq.n = n;
        b->cigar.n = n_cig; b->qual.n = want_bits ? 0 : n_qual; b->qname.n = n_name; b->qsum.n = want_bits ? n : 0;
        st.cur = o;
        t_parse += tnow() - tp0;
    }
    if (!b->cigar_off.reserve(n + 1) || !b->qual_off.reserve(n + 1) || !b->qname_off.reserve(n + 1) || !b->seq_off.reserve(n + 1) ||
        !b->pos.reserve(1) || !b->flag.reserve(1) || !b->mapq.reserve(1) || !b->cigar.reserve(1) || !b->qual.reserve(1) || !b->qname.reserve(1) ||
        !b->seq4.reserve(1)) { b->err = "out of memory"; return CL_ERR_INVALID; }
    b->cigar_off.p[n] = (uint32_t)n_cig; b->qual_off.p[n] = n_qual; b->qname_off.p[n] = (uint32_t)n_name; b->seq_off.p[n] = n_bases;
    if (timing_on())
        fprintf(stderr, "[dut-timing]   read %.0f ms, inflate %.0f ms (%s, %d threads), scan %.0f ms, parse %.0f ms\n", st.t_read * 1e3,
                st.t_inflate * 1e3, libdeflate().ok ? "libdeflate" : "zlib", dut::worker_threads(), t_scan * 1e3, t_parse * 1e3);
    b->last_tid_done = tid;
    out->n = n;
    out->pos = b->pos.p; out->flag = b->flag.p; out->mapq = b->mapq.p;
    out->cigar_off = b->cigar_off.p; out->cigar = b->cigar.p;
    out->qual_off = b->qual_off.p; out->qual = want_bits ? nullptr : b->qual.p;
    out->qname_off = b->qname_off.p; out->qname = b->qname.p;
    out->pass_bits = nullptr; out->pass_sum = nullptr;
    if (want_bits) {
        if (!b->qbits.reserve(2) || !b->qsum.reserve(1)) { b->err = "out of memory"; return CL_ERR_INVALID; }   // (a contig without records)
        if (b->qbits.n == 0) { b->qbits.p[0] = b->qbits.p[1] = 0; }
        out->pass_bits = b->qbits.p; out->pass_sum = b->qsum.p;
    }
    if (want_seq) { *seq_off = b->seq_off.p; *seq4 = b->seq4.p; }
    return CL_OK;
}

int dut_bam_read_contig(dut_bam *b, int tid, dut_records *out, const uint64_t **seq_off, const uint8_t **seq4)
{
    // no exception leaves the library through the C ABI
    try { return dut_bam_read_contig_impl(b, tid, out, seq_off, seq4); }
    catch (const std::bad_alloc &) { return CL_ERR_NOMEM; }
    catch (...) { return CL_ERR_INVALID; }
}

int dut_bam_read_contig_bits(dut_bam *b, int tid, uint8_t min_base_quality, dut_records *out)
{
    try { return dut_bam_read_contig_impl(b, tid, out, nullptr, nullptr, (int)min_base_quality); }
    catch (const std::bad_alloc &) { return CL_ERR_NOMEM; }
    catch (...) { return CL_ERR_INVALID; }
}


int dut_bam_sample(dut_bam *b, dut_bam_sample_fn fn, void *ud)
{
    if (!b || !fn) return CL_ERR_INVALID;
    b->err.clear();
    b->pending = false;
    b->last_tid_done = -1;
    b->st.valid = false;
    if (!b->z.seek(b->data_start)) { b->err = "rewind failed"; return CL_ERR_INVALID; }
    int rc = CL_OK;
    for (uint64_t i = 0;; ++i) {
        if (!read_record(b)) { if (!b->err.empty()) rc = CL_ERR_INVALID; break; }
        const uint8_t *r = b->rec.data();
        const uint32_t l_read_name = r[8];
        if (l_read_name == 0 || 32ull + l_read_name > b->rec.size()) { b->err = "malformed BAM record"; rc = CL_ERR_INVALID; break; }
        if (!fn(ud, i, rd16(r + 14), rd32(r + 16), r + 32, l_read_name - 1, (int32_t)rd32(r + 28))) break;
    }
    // leave the reader at the first record again
    b->pending = false;
    if (!b->z.seek(b->data_start) && rc == CL_OK) { b->err = "rewind failed"; rc = CL_ERR_INVALID; }
    return rc;
}

} // extern "C"

// ---------------------------------------------------------------------------------------------
// FASTA + .fai
// ---------------------------------------------------------------------------------------------
struct dut_fasta {
    FILE *fp = nullptr;
    struct Ent { std::string name; uint64_t len, off, linebases, linewidth; };
    std::vector<Ent> ents;
    std::vector<uint8_t> seq;
    std::string err;
    uint64_t file_size = 0;
};

// What `samtools faidx` / htslib's fai_build writes: one row per sequence (name up to the first white space, length,
// offset of the first base, bases per line, bytes per line).  The reference's faidx::Reader::from_path builds the
// index when it is missing; so does dut_fasta_open.
static bool fai_build(FILE *fp, std::vector<dut_fasta::Ent> &ents, std::string &err)
{
    std::vector<char> buf(1 << 20);
    uint64_t off = 0;                 // file offset of buf[0]
    dut_fasta::Ent cur{};
    bool in_seq = false, in_name_line = false, name_done = false, last_line_short = false;
    uint64_t line_b = 0, line_w = 0;  // bases / bytes of the current sequence line so far
    auto end_line = [&]() -> bool {    // a sequence line of `cur` just ended
        if (line_w == 0 && line_b == 0) return true;
        if (cur.linebases == 0) { cur.linebases = line_b; cur.linewidth = line_w; }
        else {
            if (last_line_short && line_b) { err = "different line length in sequence '" + cur.name + "'"; return false; }
            if (line_b != cur.linebases || line_w != cur.linewidth) {
                if (line_b > cur.linebases) { err = "different line length in sequence '" + cur.name + "'"; return false; }
                last_line_short = true;                       // only the last line may be shorter
            }
        }
        cur.len += line_b;
        line_b = 0; line_w = 0;
        return true;
    };
    if (fseeko(fp, 0, SEEK_SET) != 0) { err = "cannot seek"; return false; }
    for (;;) {
        const size_t got = fread(buf.data(), 1, buf.size(), fp);
        if (got == 0) break;
        for (size_t i = 0; i < got; ++i) {
            const char c = buf[i];
            if (in_name_line) {
                if (c == '\n') { in_name_line = false; cur.off = off + i + 1; }
                else if (!name_done) { if (c == ' ' || c == '\t' || c == '\r') name_done = true; else cur.name.push_back(c); }
                continue;
            }
            if (c == '>' && line_w == 0) {
                if (in_seq) { if (!end_line()) return false; ents.push_back(cur); }
                cur = dut_fasta::Ent{}; in_seq = true; in_name_line = true; name_done = false; last_line_short = false;
                continue;
            }
            if (!in_seq) { if (c == '\n' || c == '\r') continue; err = "the file does not start with '>'"; return false; }
            line_w += 1;
            if (c == '\n') { if (!end_line()) return false; }
            else if (c != '\r') line_b += 1;
        }
        off += got;
    }
    if (in_seq) {
        if (line_w) { if (cur.linebases == 0) { cur.linebases = line_b; cur.linewidth = line_w + 1; } cur.len += line_b; line_b = 0; line_w = 0; }
        ents.push_back(cur);
    }
    for (auto &e : ents) if (e.len == 0) { e.linebases = 0; e.linewidth = 0; }
    return true;
}

extern "C" {

dut_fasta *dut_fasta_open(const char *path, char *err, size_t err_len)
{
    if (!path) { set_err(err, err_len, "null path"); return nullptr; }
    dut_fasta *f = new dut_fasta();
    f->fp = fopen(path, "rb");
    if (!f->fp) { set_err(err, err_len, std::string("cannot open ") + path); delete f; return nullptr; }
    if (fseeko(f->fp, 0, SEEK_END) == 0) f->file_size = (uint64_t)ftello(f->fp);
    const std::string fai_path = std::string(path) + ".fai";
    FILE *fi = fopen(fai_path.c_str(), "r");
    if (!fi) {
        // no index beside the file: build it (and leave it there when the directory can be written, as htslib does)
        std::string berr;
        if (!fai_build(f->fp, f->ents, berr)) {
            set_err(err, err_len, std::string("cannot index ") + path + ": " + berr);
            fclose(f->fp); delete f; return nullptr;
        }
        // written under a name of this process's own and renamed into place: the ranks of a multi-GPU run all open the
        // same FASTA at once, and none of them may find a half-written index
        const std::string tmp_path = fai_path + ".tmp" + std::to_string((long long)getpid());
        if (FILE *fo = fopen(tmp_path.c_str(), "w")) {
            bool ok = true;
            for (const auto &e : f->ents)
                ok = ok && fprintf(fo, "%s\t%llu\t%llu\t%llu\t%llu\n", e.name.c_str(), (unsigned long long)e.len, (unsigned long long)e.off,
                                   (unsigned long long)e.linebases, (unsigned long long)e.linewidth) > 0;
            ok = (fclose(fo) == 0) && ok;
            if (!ok || rename(tmp_path.c_str(), fai_path.c_str()) != 0) (void)remove(tmp_path.c_str());
        }
        return f;
    }
    char line[4096];
    size_t row = 0;
    while (fgets(line, sizeof(line), fi)) {
        ++row;
        char nm[2048]; unsigned long long a, b, c, d;
        if (line[0] == '\n' || line[0] == 0) continue;
        if (sscanf(line, "%2047[^\t]\t%llu\t%llu\t%llu\t%llu", nm, &a, &b, &c, &d) != 5) {
            set_err(err, err_len, fai_path + ": malformed row " + std::to_string(row));
            fclose(fi); fclose(f->fp); delete f; return nullptr;
        }
        // a row the fetch arithmetic cannot use: an empty sequence may have no line shape, any other needs
        // 0 < bases per line <= bytes per line
        const bool ok = a == 0 || (c > 0 && d >= c);
        if (!ok) {
            set_err(err, err_len, fai_path + ": invalid row " + std::to_string(row) + " (sequence '" + nm + "')");
            fclose(fi); fclose(f->fp); delete f; return nullptr;
        }
        f->ents.push_back({nm, a, b, c, d});
    }
    fclose(fi);
    return f;
}

void dut_fasta_close(dut_fasta *f)
{
    if (!f) return;
    if (f->fp) fclose(f->fp);
    delete f;
}

const char *dut_fasta_error(const dut_fasta *f) { return f ? f->err.c_str() : "null reader"; }

static int dut_fasta_fetch_impl(dut_fasta *f, const char *name, const uint8_t **bases, uint64_t *len)
{
    for (const auto &e : f->ents) {
        if (e.name != name) continue;
        f->seq.clear();
        if (e.len == 0) return CL_OK;
        const uint64_t n_lines = (e.len + e.linebases - 1) / e.linebases;
        const uint64_t span = e.len + (n_lines - 1) * (e.linewidth - e.linebases);
        // a file shorter than its index says: the bases that are there, the rest reads as 'N' (mod.rs:79-80)
        const uint64_t want = f->file_size ? std::min<uint64_t>(span, f->file_size > e.off ? f->file_size - e.off : 0) : span;
        std::vector<uint8_t> raw(want);
        if (fseeko(f->fp, (off_t)e.off, SEEK_SET) != 0) { f->err = std::string("cannot seek to sequence '") + name + "'"; return CL_ERR_INVALID; }
        const size_t got = fread(raw.data(), 1, want, f->fp);
        if (got < want && ferror(f->fp)) { f->err = std::string("read error in sequence '") + name + "'"; return CL_ERR_INVALID; }
        f->seq.reserve(e.len);
        for (size_t i = 0; i < got && f->seq.size() < e.len; ++i) {
            if (i % e.linewidth < e.linebases) f->seq.push_back(raw[i]);
        }
        *bases = f->seq.data(); *len = f->seq.size();
        return CL_OK;
    }
    // faidx fetch_seq of a name the index does not hold is an error in the reference (mod.rs:79: `?`), not a run of 'N'
    f->err = std::string("sequence '") + name + "' not found in the reference FASTA index";
    return CL_ERR_INVALID;
}

int dut_fasta_fetch(dut_fasta *f, const char *name, const uint8_t **bases, uint64_t *len)
{
    if (!f || !name || !bases || !len) return CL_ERR_INVALID;
    *bases = nullptr; *len = 0;
    f->err.clear();
    // no exception leaves the library through the C ABI (and none may end a helper thread)
    try { return dut_fasta_fetch_impl(f, name, bases, len); }
    catch (const std::bad_alloc &) { f->err = "out of memory"; return CL_ERR_NOMEM; }
    catch (...) { f->err = "internal error"; return CL_ERR_INVALID; }
}

} // extern "C"

// ---------------------------------------------------------------------------------------------
// the file-level coverage driver
// ---------------------------------------------------------------------------------------------
static double io_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void io_stage_time(const char *what, double &t0)
{
    static const bool on = getenv("DUT_TIMING") && *getenv("DUT_TIMING") == '1';
    if (!on) return;
    const double t1 = io_now();
    fprintf(stderr, "[dut-timing] %-28s %8.1f ms\n", what, (t1 - t0) * 1e3);
    t0 = t1;
}

// Contigs dealt to devices by longest-processing-time-first: the heaviest contig next, to the device with the least
// load so far (weights: the index's mapped-read counts when it records them, the contig lengths otherwise).
static std::vector<int> lpt_deal(const std::vector<uint64_t> &weight, size_t n_dev)
{
    std::vector<size_t> order(weight.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return weight[a] > weight[b]; });
    std::vector<uint64_t> load(n_dev, 0);
    std::vector<int> owner(weight.size(), 0);
    for (size_t i : order) {
        size_t best = 0;
        for (size_t d = 1; d < n_dev; ++d) if (load[d] < load[best]) best = d;
        owner[i] = (int)best; load[best] += weight[i] + 1;
    }
    return owner;
}

static int dut_coverage_files_impl(const char *bam_path, const char *fasta_path, const char *bed_path,
                                  const char *summary_json, const char *summary_html, const cl_options *opt,
                                  const char *const *contigs, size_t n_contigs, const int *devices, size_t n_devices,
                                  unsigned flags, char *err, size_t err_len)
{
    if (!bam_path || !fasta_path || !bed_path || !opt || !devices || n_devices == 0) { set_err(err, err_len, "null argument"); return CL_ERR_INVALID; }
    const int device_id = devices[0];
    const bool leave = (flags & DUT_FILES_LEAVE_TO_EXIT) != 0;
    // the engine's default form wants one bit per base (cl_create reads the same variable): the reader then takes the
    // base-quality test while it parses the records and no quality byte leaves it
    const char *qf = getenv("DUT_QUAL_FORM");
    const char *pr = getenv("DUT_PACKED_READER");                   // =0: bytes from the reader, tested in cl_push_reads (for A/B timing)
    const bool use_bits = !(qf && strcmp(qf, "bytes") == 0) && !(pr && pr[0] == '0');
    char e[512] = {0};
    double tm = io_now();
    dut_bam_stats *bstats = dut_bam_stats_new(10000);                  // api/coverage.rs:56-59
    if (dut_bam_stats_collect(bstats, bam_path, e, sizeof(e)) != CL_OK) {
        set_err(err, err_len, std::string("Failed to collect BAM stats: ") + e);
        dut_bam_stats_free(bstats);
        return CL_ERR_INVALID;
    }
    dut_bam *bam = dut_bam_open(bam_path, e, sizeof(e));
    if (!bam) { dut_bam_stats_free(bstats); set_err(err, err_len, std::string("Failed to open BAM file: ") + e); return CL_ERR_INVALID; }   // api/coverage.rs:69-70
    dut_fasta *fa = dut_fasta_open(fasta_path, e, sizeof(e));
    if (!fa) { dut_bam_stats_free(bstats); dut_bam_close(bam); set_err(err, err_len, std::string("Failed to open reference: ") + e); return CL_ERR_INVALID; }   // :73-74
    // initialize_contig_stats / validate_contig_selection, api/coverage.rs:149-204
    std::vector<int> tids;
    for (int t = 0; t < dut_bam_n_ref(bam); ++t) {
        bool take = contigs == nullptr;
        for (size_t i = 0; !take && i < n_contigs; ++i) take = strcmp(contigs[i], dut_bam_ref_name(bam, t)) == 0;
        if (take) tids.push_back(t);
    }
    int rc = CL_OK;
    cl_ctx *ctx = nullptr;
    dut_profiler *prof = nullptr;
    std::vector<dut_contig_stats> stats;
    std::vector<std::string> names;
    std::vector<std::vector<uint64_t>> counts;
    if (contigs && tids.empty()) {
        std::string list;
        for (size_t i = 0; i < n_contigs; ++i) { if (i) list += ", "; list += contigs[i]; }
        set_err(err, err_len, "None of the specified contigs (" + list + ") were found in the BAM file");
        rc = CL_ERR_INVALID; goto out;
    }
    if (n_devices > 1 && !tids.empty()) {
        // ---- several devices: one host thread, one reader pair and one engine context per device; the contigs dealt by
        //      LPT; every contig's runs, counts and statistics come back through host memory and the BED is written here,
        //      in tid order (api/coverage.rs:229-234 is a serial loop with no cross-contig state but the BED writer's
        //      pending line, callable_profiler.rs:64-66).  No collective: one process holds every result. ----
        io_stage_time("(before contigs)", tm);
        struct Result { int rc = CL_OK; std::string msg; dut_contig_stats st; uint64_t counts[6]; std::vector<cl_interval> iv; bool done = false; };
        std::vector<Result> res(tids.size());
        std::mutex mu;
        std::condition_variable cv;
        std::atomic<bool> stop{false};
        std::vector<uint64_t> weight(tids.size());
        for (size_t i = 0; i < tids.size(); ++i) {
            const int64_t m = dut_bam_ref_mapped(bam, tids[i]);
            weight[i] = m >= 0 ? (uint64_t)m : (uint64_t)dut_bam_ref_len(bam, tids[i]);
        }
        const std::vector<int> owner = lpt_deal(weight, n_devices);
        prof = dut_profiler_new(bed_path);
        if (!prof) { set_err(err, err_len, std::string("Failed to create CallableProfiler: cannot create ") + bed_path); rc = CL_ERR_INVALID; goto out; }
        {
            uint32_t largest = 0;
            for (int t : tids) if (strcmp(dut_bam_ref_name(bam, t), "chrM") != 0) largest = std::max(largest, dut_bam_ref_len(bam, t));
            dut_profiler_enable_plots(prof, largest);
        }
        {
            std::vector<dut::Thread> workers;
            for (size_t d = 0; d < n_devices; ++d) {
                workers.push_back(dut::spawn_or_run([&, d]() {
                    auto fail_rest = [&](size_t from, int code, const std::string &m) {
                        std::lock_guard<std::mutex> g(mu);
                        for (size_t i = from; i < tids.size(); ++i)
                            if (owner[i] == (int)d && !res[i].done) { res[i].rc = code; res[i].msg = m; res[i].done = true; }
                        cv.notify_all();
                    };
                    char e2[512] = {0};
                    cl_ctx *dctx = nullptr;
                    dut_bam *db = nullptr; dut_fasta *df = nullptr;
                    try {
                        int drc = cl_create(opt, devices[d], nullptr, &dctx);
                        if (drc != CL_OK) { fail_rest(0, drc, "no usable HIP device (the engine has no CPU fallback)"); return; }
                        db = dut_bam_open(bam_path, e2, sizeof(e2));
                        if (!db) { fail_rest(0, CL_ERR_INVALID, std::string("Failed to open BAM file: ") + e2); cl_destroy(dctx); return; }
                        df = dut_fasta_open(fasta_path, e2, sizeof(e2));
                        if (!df) { fail_rest(0, CL_ERR_INVALID, std::string("Failed to open reference: ") + e2); dut_bam_close(db); cl_destroy(dctx); return; }
                        for (size_t i = 0; i < tids.size(); ++i) {
                            if (owner[i] != (int)d) continue;
                            if (stop.load()) { fail_rest(i, CL_ERR_INVALID, "abandoned"); break; }
                            const int t = tids[i];
                            dut_records rec;
                            const uint8_t *bases = nullptr; uint64_t blen = 0;
                            int frc = CL_OK;
                            dut::Thread fb;
                            if (dut_bam_ref_len(db, t) > 0)
                                fb = dut::spawn_or_run([&]() { frc = dut_fasta_fetch(df, dut_bam_ref_name(db, t), &bases, &blen); });
                            drc = use_bits ? dut_bam_read_contig_bits(db, t, opt->min_base_quality, &rec) : dut_bam_read_contig(db, t, &rec, nullptr, nullptr);
                            if (fb.joinable()) fb.join();
                            if (drc != CL_OK) { fail_rest(i, drc, std::string("Error processing contig: ") + dut_bam_error(db)); break; }
                            if (frc != CL_OK) { fail_rest(i, frc, std::string("Error processing contig: ") + dut_fasta_error(df)); break; }
                            Result r;
                            memset(&r.st, 0, sizeof(r.st));
                            const cl_interval *iv = nullptr; size_t niv = 0;
                            drc = dut_process_single_contig_runs(dctx, &r.st, opt, t, dut_bam_ref_len(db, t), bases, blen, &rec, r.counts, &iv, &niv);
                            if (drc != CL_OK) {
                                const char *m = cl_last_error(dctx);
                                fail_rest(i, drc, std::string("Error processing contig: ") + ((m && *m) ? m : (drc == CL_ERR_UNSORTED ? "the input is not sorted" : "failed")));
                                break;
                            }
                            r.iv.assign(iv, iv + niv);
                            r.done = true;
                            { std::lock_guard<std::mutex> g(mu); res[i] = std::move(r); }
                            cv.notify_all();
                        }
                    } catch (...) { fail_rest(0, CL_ERR_NOMEM, "out of memory or internal error"); }
                    if (!leave) {
                        if (df) dut_fasta_close(df);
                        if (db) dut_bam_close(db);
                        if (dctx) cl_destroy(dctx);
                    }
                }));
            }
            // the BED, in tid order, as each contig's result arrives
            for (size_t i = 0; i < tids.size(); ++i) {
                Result r;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return res[i].done; });
                    r = std::move(res[i]);
                }
                if (rc != CL_OK) continue;                       // (after the first error the rest is only waited for)
                if (r.rc != CL_OK) { rc = r.rc; set_err(err, err_len, r.msg); stop.store(true); continue; }
                const char *nm = dut_bam_ref_name(bam, tids[i]);
                double tb = io_now();
                int frc = dut_profiler_feed_contig(prof, nm, r.iv.data(), r.iv.size(), r.counts);
                if (frc == CL_OK && dut_profiler_finish_plot(prof, nm, dut_bam_ref_len(bam, tids[i])) < 0) frc = CL_ERR_INVALID;
                if (frc != CL_OK) { rc = frc; set_err(err, err_len, std::string("cannot write ") + bed_path); stop.store(true); continue; }
                io_stage_time("BED lines", tb);
                stats.push_back(r.st); names.push_back(nm); counts.push_back(std::vector<uint64_t>(r.counts, r.counts + 6));
            }
            workers.clear();                                     // joins
        }
        io_stage_time("contigs over the devices", tm);
        if (rc != CL_OK) goto out;
    } else {
        // the HIP runtime and the engine context come up on their own thread while the first contig is decoded
        dut::Thread init = dut::spawn_or_run([&]() { rc = cl_create(opt, device_id, nullptr, &ctx); });
        // Contigs are processed in ascending tid order (api/coverage.rs:229-234).  With an index and more
        // than one contig, the records and reference bases of contig i+1 are read by a second reader on
        // its own thread while contig i is admitted, pushed, run and written (DUT_PIPELINE=0: off).
        struct Slot { dut_bam *bam = nullptr; dut_fasta *fa = nullptr; dut_records rec; const uint8_t *bases = nullptr; uint64_t blen = 0; int rc = CL_OK, frc = CL_OK; };
        Slot slot[2];
        slot[0].bam = bam; slot[0].fa = fa;
        const char *pe = getenv("DUT_PIPELINE");
        bool pipeline = tids.size() > 1 && dut_bam_has_index(bam) && !(pe && *pe == '0');
        if (pipeline) {
            slot[1].bam = dut_bam_open(bam_path, e, sizeof(e));
            slot[1].fa = slot[1].bam ? dut_fasta_open(fasta_path, e, sizeof(e)) : nullptr;
            if (!slot[1].bam || !slot[1].fa) {                 // e.g. out of file handles: read in line instead
                if (slot[1].bam) dut_bam_close(slot[1].bam);
                slot[1].bam = nullptr; pipeline = false;
            }
        }
        auto fetch = [&](Slot &s, int t) {
            // the reference bases (one thread: read + strip the line ends) beside the record decode (all threads)
            s.bases = nullptr; s.blen = 0;
            // (a zero-length contig fetches nothing: the reference's loops over it run zero times, mod.rs:65-147, so a
            // FASTA that lacks such an @SQ is not an error)
            s.frc = CL_OK;
            dut::Thread fb;
            if (dut_bam_ref_len(s.bam, t) > 0)
                fb = dut::spawn_or_run([&]() { s.frc = dut_fasta_fetch(s.fa, dut_bam_ref_name(s.bam, t), &s.bases, &s.blen); });
            s.rc = use_bits ? dut_bam_read_contig_bits(s.bam, t, opt->min_base_quality, &s.rec) : dut_bam_read_contig(s.bam, t, &s.rec, nullptr, nullptr);
            if (fb.joinable()) fb.join();
        };
        dut::Thread ahead;
        io_stage_time("(before contigs)", tm);
        if (!tids.empty()) fetch(slot[0], tids[0]);
        if (init.joinable()) init.join();
        if (rc != CL_OK) set_err(err, err_len, "no usable HIP device (the engine has no CPU fallback)");
        else {
            prof = dut_profiler_new(bed_path);
            if (!prof) { set_err(err, err_len, std::string("Failed to create CallableProfiler: cannot create ") + bed_path); rc = CL_ERR_INVALID; }
            else {
                uint32_t largest = 0;                       // api/coverage.rs:210-215: the longest selected contig but chrM
                for (int t : tids) if (strcmp(dut_bam_ref_name(bam, t), "chrM") != 0) largest = std::max(largest, dut_bam_ref_len(bam, t));
                dut_profiler_enable_plots(prof, largest);
            }
        }
        for (size_t i = 0; rc == CL_OK && i < tids.size(); ++i) {
            const int t = tids[i];
            Slot &cur = pipeline ? slot[i & 1] : slot[0];
            if (i > 0 && !pipeline) fetch(cur, t);
            io_stage_time(pipeline && i > 0 ? "wait for the read-ahead" : "BAM read + decode, FASTA fetch", tm);
            if (pipeline && i + 1 < tids.size()) { Slot &nx = slot[(i + 1) & 1]; const int tn = tids[i + 1]; ahead = dut::spawn_or_run([&fetch, &nx, tn]() { fetch(nx, tn); }); }
            if (cur.rc != CL_OK) { set_err(err, err_len, std::string("Error processing contig: ") + dut_bam_error(cur.bam)); rc = cur.rc; }
            else if (cur.frc != CL_OK) { set_err(err, err_len, std::string("Error processing contig: ") + dut_fasta_error(cur.fa)); rc = cur.frc; }   // fetch_seq(..)?, mod.rs:79
            dut_contig_stats st;
            memset(&st, 0, sizeof(st));
            if (rc == CL_OK) {
                rc = dut_process_single_contig(ctx, prof, &st, opt, dut_bam_ref_name(bam, t), t, dut_bam_ref_len(bam, t), cur.bases, cur.blen, &cur.rec);
                if (rc != CL_OK) {
                    const char *m = cl_last_error(ctx);
                    set_err(err, err_len, std::string("Error processing contig: ") + ((m && *m) ? m : (rc == CL_ERR_UNSORTED ? "the input is not sorted" : "failed")));
                }
            }
            if (ahead.joinable()) ahead.join();
            if (rc != CL_OK) break;
            uint64_t c6[6];
            dut_profiler_contig_counts(prof, dut_bam_ref_name(bam, t), c6);
            stats.push_back(st); names.push_back(dut_bam_ref_name(bam, t)); counts.push_back(std::vector<uint64_t>(c6, c6 + 6));
        }
        if (slot[1].bam) { dut_fasta_close(slot[1].fa); dut_bam_close(slot[1].bam); }
        if (rc != CL_OK) goto out;
    }
    if (summary_json || summary_html) {
        std::vector<const char *> nm;
        std::vector<uint64_t> c6;
        for (size_t i = 0; i < stats.size(); ++i) { nm.push_back(names[i].c_str()); c6.insert(c6.end(), counts[i].begin(), counts[i].end()); }
        // collect_coverage_plots (api/coverage.rs:263-274): the figures that exist relative to the working directory
        // (they are written beside the BED file); listed in tid order here, in HashMap order there
        std::vector<std::string> plots;
        for (size_t i = 0; i < stats.size(); ++i) {
            const std::string pth = names[i] + "_coverage.svg";
            if (FILE *pf = fopen(pth.c_str(), "rb")) { fclose(pf); plots.push_back(pth); }
        }
        std::vector<const char *> plot_ptrs;
        for (const std::string &q : plots) plot_ptrs.push_back(q.c_str());
        dut_export_meta meta;
        memset(&meta, 0, sizeof(meta));
        meta.aligner = dut_bam_stats_aligner(bstats);
        meta.reference_build = dut_bam_stats_reference_build(bstats);
        meta.sequencing_platform = dut_bam_stats_infer_platform(bstats);
        meta.read_length = dut_bam_stats_average_read_length(bstats);
        meta.bed_file = bed_path;
        meta.summary_html = summary_html ? summary_html : "summary.html";
        meta.coverage_plots = plot_ptrs.data(); meta.n_coverage_plots = plot_ptrs.size();
        if (summary_html) {                                   // api/coverage.rs:104
            rc = dut_write_html_report(stats.data(), nm.data(), c6.data(), stats.size(), &meta, 10000, summary_html);
            if (rc != CL_OK) { set_err(err, err_len, std::string("cannot create ") + summary_html); goto out; }
        }
    if (summary_json) {
        // CoverageOutput as main.rs:68-69 serialises it
        char *js = nullptr; size_t jl = 0;
        rc = dut_coverage_output_json(stats.data(), nm.data(), c6.data(), stats.size(), &meta, &js, &jl);
        if (rc != CL_OK) { set_err(err, err_len, "cannot build the summary"); goto out; }
        FILE *jf = fopen(summary_json, "wb");
        if (!jf) { dut_free(js); set_err(err, err_len, std::string("cannot create ") + summary_json); rc = CL_ERR_INVALID; goto out; }
        fwrite(js, 1, jl, jf);
        fclose(jf);
        dut_free(js);
    }
    }
out:
    io_stage_time("(since the last decode) + summary", tm);
    dut_bam_stats_free(bstats);
    if (prof) dut_profiler_free(prof);
    if (leave) {
        // the caller is about to leave the process (DUT_FILES_LEAVE_TO_EXIT: the command line tool): every result is on
        // disk; the device context, the readers and their decode buffers are left to the exit -- giving them back one by
        // one costs a few hundred milliseconds of page-table and driver work that the exit does once, in one sweep
        if (bam && bam->z.fp) { bam->st.drop_ahead(); fclose(bam->z.fp); bam->z.fp = nullptr; }
        io_stage_time("left to the exit", tm);
        return rc;
    }
    {
        // giving the device memory back and unmapping the decode buffers take a few hundred ms at chr21 size: side by
        // side, and both joined -- a library call leaves no thread behind
        dut::Thread td = dut::spawn_or_run([&]() { if (ctx) cl_destroy(ctx); });
        dut_fasta_close(fa);
        if (bam) dut_bam_close(bam);
        io_stage_time("readers closed", tm);
        if (td.joinable()) td.join();
    }
    io_stage_time("engine destroyed", tm);
    return rc;
}

extern "C" int dut_coverage_files(const char *bam_path, const char *fasta_path, const char *bed_path,
                                  const char *summary_json, const char *summary_html, const cl_options *opt,
                                  const char *const *contigs, size_t n_contigs, int device_id, char *err, size_t err_len)
{
    // no exception leaves the library through the C ABI
    try { return dut_coverage_files_impl(bam_path, fasta_path, bed_path, summary_json, summary_html, opt, contigs, n_contigs, &device_id, 1, 0u, err, err_len); }
    catch (const std::bad_alloc &) { set_err(err, err_len, "out of memory or internal error"); return CL_ERR_NOMEM; }
    catch (...) { set_err(err, err_len, "out of memory or internal error"); return CL_ERR_INVALID; }
}

extern "C" int dut_coverage_files_multi(const char *bam_path, const char *fasta_path, const char *bed_path,
                                        const char *summary_json, const char *summary_html, const cl_options *opt,
                                        const char *const *contigs, size_t n_contigs, const int *devices, size_t n_devices,
                                        unsigned flags, char *err, size_t err_len)
{
    try { return dut_coverage_files_impl(bam_path, fasta_path, bed_path, summary_json, summary_html, opt, contigs, n_contigs, devices, n_devices, flags, err, err_len); }
    catch (const std::bad_alloc &) { set_err(err, err_len, "out of memory or internal error"); return CL_ERR_NOMEM; }
    catch (...) { set_err(err, err_len, "out of memory or internal error"); return CL_ERR_INVALID; }
}

