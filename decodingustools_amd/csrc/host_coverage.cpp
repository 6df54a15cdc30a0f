// host_coverage.cpp -- implementation of include/dut_coverage.h: the host-side part of the
// `coverage` path (read admission, BED writer, per-contig driver, derived statistics).
// The per-position work is done by the device engine (callable_loci.hip); nothing here computes
// depths or states.
#include "../../include/dut_coverage.h"
#include "host_parallel.h"

#include <algorithm>
#include <array>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <queue>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

const char *const kStateNames[6] = {"REF_N", "CALLABLE", "NO_COVERAGE", "LOW_COVERAGE",
                                    "EXCESSIVE_COVERAGE", "POOR_MAPPING_QUALITY"};

inline bool ref_consuming(uint32_t op) { return op == 0 || op == 2 || op == 3 || op == 7 || op == 8; }

uint64_t ref_length(const uint32_t *cig, uint32_t n)
{
    uint64_t l = 0;
    for (uint32_t k = 0; k < n; ++k)
        if (ref_consuming(cig[k] & 15u)) l += cig[k] >> 4;
    return l;
}

// exact set of byte strings (HashSet<Vec<u8>>, contig_profiler.rs:20,59-62): open addressing on a
// 64-bit hash, full comparison on hash match
struct NameSet {
    std::vector<uint64_t> hash;
    std::vector<uint64_t> idx;
    uint64_t count = 0;
    const dut_records *rec = nullptr;

    static uint64_t mix(const uint8_t *s, uint32_t n)
    {
        uint64_t h = 0x9E3779B97F4A7C15ull ^ n;
        uint32_t i = 0;
        for (; i + 8 <= n; i += 8) {
            uint64_t v; memcpy(&v, s + i, 8);
            h = (h ^ v) * 0xff51afd7ed558ccdull; h ^= h >> 32;
        }
        uint64_t v = 0;
        if (i < n) memcpy(&v, s + i, n - i);
        h = (h ^ v) * 0xc4ceb9fe1a85ec53ull; h ^= h >> 29;
        h *= 0xff51afd7ed558ccdull; h ^= h >> 32;
        return h ? h : 1;
    }
    void grow()
    {
        const size_t ncap = hash.empty() ? 4096 : hash.size() * 2;
        std::vector<uint64_t> nh(ncap, 0), ni(ncap, 0);
        for (size_t i = 0; i < hash.size(); ++i)
            if (hash[i]) {
                size_t j = hash[i] & (ncap - 1);
                while (nh[j]) j = (j + 1) & (ncap - 1);
                nh[j] = hash[i]; ni[j] = idx[i];
            }
        hash.swap(nh); idx.swap(ni);
    }
    uint64_t hash_of(uint64_t r) const { return mix(rec->qname + rec->qname_off[r], rec->qname_off[r + 1] - rec->qname_off[r]); }
    void presize(uint64_t expected)
    {
        size_t cap = 4096;
        while (cap < expected * 2 + 2) cap *= 2;
        hash.assign(cap, 0); idx.assign(cap, 0); count = 0;
    }
    bool insert(uint64_t r) { return insert_hashed(r, hash_of(r)); }
    bool insert_hashed(uint64_t r, uint64_t h)
    {
        if ((count + 1) * 2 > hash.size()) grow();
        const uint8_t *nm = rec->qname + rec->qname_off[r];
        const uint32_t nl = rec->qname_off[r + 1] - rec->qname_off[r];
        size_t j = h & (hash.size() - 1);
        while (hash[j]) {
            if (hash[j] == h) {
                const uint64_t o = idx[j];
                const uint32_t ol = rec->qname_off[o + 1] - rec->qname_off[o];
                if (ol == nl && memcmp(rec->qname + rec->qname_off[o], nm, nl) == 0) return false;
            }
            j = (j + 1) & (hash.size() - 1);
        }
        hash[j] = h; idx[j] = r; ++count;
        return true;
    }
};

} // namespace

struct dut_profiler {
    FILE *bed = nullptr;
    bool has_state = false;
    std::string cur_contig;
    uint64_t cur_start = 0, cur_end = 0;
    uint32_t cur_state = 0;
    std::unordered_map<std::string, std::array<uint64_t, 6>> counts;    // contig_counts, callable_profiler.rs:13
    // coverage figure (callable_profiler.rs:48-59, 64-84): every written line of one of the three plotted states
    // is also a range of the current figure; finish_plot bins and clears them
    struct PlotRange { uint32_t start, end, state; };
    bool plots = false;
    uint32_t largest_contig_length = 0;
    std::string out_dir;
    std::vector<PlotRange> ranges;

    // The BED text is assembled in memory (a contig has hundreds of thousands of lines: one formatted-print call per
    // line was a third of a contig's host time) and handed to the file a megabyte at a time.
    std::string text;
    static void put_u64(std::string &o, uint64_t v)
    {
        char d[20];
        int n = 0;
        do { d[n++] = (char)('0' + v % 10); v /= 10; } while (v);
        while (n) o.push_back(d[--n]);
    }
    void flush_text()
    {
        if (!text.empty()) { fwrite(text.data(), 1, text.size(), bed); text.clear(); }
    }
    void write_state()                       // callable_profiler.rs:39-62
    {
        if (!has_state) return;
        text += cur_contig; text.push_back('\t');
        put_u64(text, cur_start); text.push_back('\t');
        put_u64(text, cur_end); text.push_back('\t');
        text += kStateNames[cur_state]; text.push_back('\n');
        if (text.size() >= (1u << 20)) flush_text();
        if (plots && (cur_state == CL_CALLABLE || cur_state == CL_POOR_MAPPING_QUALITY || cur_state == CL_REF_N))
            ranges.push_back({(uint32_t)cur_start, (uint32_t)cur_end, cur_state});      // `as u32`, :53-54
    }
};

namespace {
// histogram_plotter.rs:74-101 + 412-440: positions of the three plotted states per stride of the contig.
// stride: ceil(16569 / 200) for "chrM", else ceil(largest selected non-chrM contig / 2000).
bool plot_bins(const dut_profiler *p, const char *contig, uint32_t contig_length, uint32_t &stride,
               std::vector<uint32_t> &call, std::vector<uint32_t> &lowq, std::vector<uint32_t> &refn)
{
    stride = strcmp(contig, "chrM") == 0 ? (16569u + 200u - 1u) / 200u : (uint32_t)(((uint64_t)p->largest_contig_length + 2000u - 1u) / 2000u);
    if (stride == 0) return false;                                        // the reference divides by zero here
    const size_t n = (size_t)(contig_length / stride) + 1;
    call.assign(n, 0); lowq.assign(n, 0); refn.assign(n, 0);
    for (const dut_profiler::PlotRange &r : p->ranges) {
        std::vector<uint32_t> &dst = r.state == CL_CALLABLE ? call : (r.state == CL_REF_N ? refn : lowq);
        for (uint64_t pos = r.start; pos < r.end;) {                      // bin by bin instead of position by position
            const uint64_t idx = pos / stride, stop = std::min<uint64_t>(r.end, (idx + 1) * stride);
            if (idx < n) dst[idx] += (uint32_t)(stop - pos);              // ranges past the arrays are ignored (:88)
            pos = stop;
        }
    }
    return true;
}

std::string xml_escape(const std::string &s)
{
    std::string o;
    for (char c : s) {
        switch (c) {
        case '&': o += "&amp;"; break; case '<': o += "&lt;"; break; case '>': o += "&gt;"; break;
        case '"': o += "&quot;"; break; case '\'': o += "&apos;"; break; default: o += c;
        }
    }
    return o;
}

// The figure itself is this project's own drawing of those three arrays (the reference's SVG markup is
// presentation and is not reproduced): one 1-px column per stride, stacked callable / poor mapping quality /
// reference N as fractions of the stride, equal neighbouring columns merged into one rectangle.
std::string plot_svg(const std::string &contig, uint32_t contig_length, uint32_t stride, const std::vector<uint32_t> &call,
                     const std::vector<uint32_t> &lowq, const std::vector<uint32_t> &refn)
{
    const int H = 100, top = 24, left = 8;
    const size_t n = call.size();
    char buf[256];
    std::string o;
    snprintf(buf, sizeof(buf), "<svg xmlns=\"http://www.w3.org/2000/svg\" width=\"%zu\" height=\"%d\" role=\"img\">\n", n + 2 * left, H + top + 22);
    o += buf;
    o += "<title>Coverage distribution for " + xml_escape(contig) + "</title>\n";
    snprintf(buf, sizeof(buf), "<rect x=\"0\" y=\"0\" width=\"%zu\" height=\"%d\" fill=\"#ffffff\"/>\n", n + 2 * left, H + top + 22);
    o += buf;
    o += "<text x=\"8\" y=\"15\" font-family=\"sans-serif\" font-size=\"12\">" + xml_escape(contig);
    snprintf(buf, sizeof(buf), " (%u bp, %u bp per column): callable green, poor mapping quality orange, reference N grey</text>\n", contig_length, stride);
    o += buf;
    auto px = [&](uint32_t v) { return (int)(((uint64_t)std::min(v, stride) * H + stride / 2) / stride); };
    const char *fill[3] = {"#2e8b57", "#e69f00", "#9e9e9e"};
    const std::vector<uint32_t> *arr[3] = {&call, &lowq, &refn};
    for (size_t a = 0; a < n;) {
        size_t b = a + 1;
        while (b < n && px(call[b]) == px(call[a]) && px(lowq[b]) == px(lowq[a]) && px(refn[b]) == px(refn[a])) ++b;
        int y = top + H;
        for (int k = 0; k < 3; ++k) {
            const int h = px((*arr[k])[a]);
            if (h > 0) {
                y -= h;
                snprintf(buf, sizeof(buf), "<rect x=\"%zu\" y=\"%d\" width=\"%zu\" height=\"%d\" fill=\"%s\"/>\n", left + a, y < top ? top : y, b - a, h, fill[k]);
                o += buf;
            }
        }
        a = b;
    }
    snprintf(buf, sizeof(buf), "<line x1=\"%d\" y1=\"%d\" x2=\"%zu\" y2=\"%d\" stroke=\"#000000\" stroke-width=\"1\"/>\n", left, top + H, left + n, top + H);
    o += buf;
    snprintf(buf, sizeof(buf), "<text x=\"%d\" y=\"%d\" font-family=\"sans-serif\" font-size=\"10\">0</text>\n", left, top + H + 14);
    o += buf;
    snprintf(buf, sizeof(buf), "<text x=\"%zu\" y=\"%d\" font-family=\"sans-serif\" font-size=\"10\" text-anchor=\"end\">%u</text>\n", left + n, top + H + 14, contig_length);
    o += buf;
    o += "</svg>\n";
    return o;
}
} // namespace

extern "C" {

const char *dut_state_name(uint32_t s) { return s < 6 ? kStateNames[s] : "?"; }

dut_profiler *dut_profiler_new(const char *bed_path)
{
    if (!bed_path) return nullptr;
    FILE *f = fopen(bed_path, "wb");                       // File::create, :31
    if (!f) return nullptr;
    setvbuf(f, nullptr, _IOFBF, 1 << 20);
    dut_profiler *p = new dut_profiler();
    p->bed = f;
    const std::string bp(bed_path);                        // output_dir = parent of the BED file, :23-26
    const size_t slash = bp.find_last_of('/');
    p->out_dir = slash == std::string::npos ? std::string() : (slash == 0 ? std::string("/") : bp.substr(0, slash));
    return p;
}

void dut_profiler_enable_plots(dut_profiler *p, uint32_t largest_contig_length)
{
    if (!p) return;
    p->plots = true; p->largest_contig_length = largest_contig_length;
}

int dut_profiler_plot_bins(const dut_profiler *p, const char *contig, uint32_t contig_length, uint32_t *stride,
                           uint32_t *callable, uint32_t *low_qual, uint32_t *ref_n, size_t cap, size_t *n_bins)
{
    if (!p || !contig || !n_bins) return CL_ERR_INVALID;
    std::vector<uint32_t> c, l, r;
    uint32_t st = 0;
    if (!plot_bins(p, contig, contig_length, st, c, l, r)) return CL_ERR_INVALID;
    *n_bins = c.size();
    if (stride) *stride = st;
    if (cap >= c.size()) {
        if (callable) memcpy(callable, c.data(), c.size() * 4);
        if (low_qual) memcpy(low_qual, l.data(), l.size() * 4);
        if (ref_n) memcpy(ref_n, r.data(), r.size() * 4);
    }
    return CL_OK;
}

int dut_profiler_finish_plot(dut_profiler *p, const char *contig, uint32_t contig_length)
{
    if (!p || !contig) return CL_ERR_INVALID;
    if (!p->plots || p->ranges.empty()) return 0;          // finish_contig: nothing to draw, :67
    std::vector<uint32_t> c, l, r;
    uint32_t st = 0;
    if (!plot_bins(p, contig, contig_length, st, c, l, r)) { p->ranges.clear(); return CL_ERR_INVALID; }
    const std::string svg = plot_svg(contig, contig_length, st, c, l, r);
    p->ranges.clear();                                     // std::mem::take, :69
    const std::string path = (p->out_dir.empty() ? std::string() : (p->out_dir == "/" ? p->out_dir : p->out_dir + "/")) + contig + "_coverage.svg";
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return CL_ERR_INVALID;
    fwrite(svg.data(), 1, svg.size(), f);
    fclose(f);
    return 1;
}

void dut_profiler_free(dut_profiler *p)
{
    if (!p) return;
    if (p->bed) { p->flush_text(); fclose(p->bed); }
    delete p;
}

void dut_profiler_contig_counts(const dut_profiler *p, const char *contig, uint64_t out[6])
{
    for (int i = 0; i < 6; ++i) out[i] = 0;
    if (!p || !contig) return;
    const auto it = p->counts.find(contig);
    if (it != p->counts.end()) for (int i = 0; i < 6; ++i) out[i] = it->second[i];
}

int dut_profiler_feed_contig(dut_profiler *p, const char *contig, const cl_interval *iv, size_t n_iv,
                             const uint64_t state_counts[6])
{
    if (!p || !contig || (!iv && n_iv)) return CL_ERR_INVALID;
    // contig_counts[contig][state] += 1 per position (:124-126): the device counted them
    if (n_iv) {
        std::array<uint64_t, 6> &slot = p->counts[contig];      // value-initialised to zeros on first use
        for (int i = 0; i < 6; ++i) slot[i] += state_counts ? state_counts[i] : 0;
    }
    // The engine's intervals are maximal runs: neighbours differ in state.  Then, once the first interval has gone
    // through the rules below, every further one just writes its predecessor and takes its place -- line i of the
    // contig is interval i, whatever came before -- and the text of a contig with hundreds of thousands of lines is
    // formatted in chunks on all host threads (it was a tenth of a contig's host time on one).  Anything else (equal
    // neighbours, a malformed interval, the plot ranges wanted) takes the sequential loop.
    size_t n_seq = n_iv;
    if (n_iv >= 4096 && !p->plots) {
        const size_t grain = 16384, nchunk = (n_iv + grain - 1) / grain;
        std::vector<uint8_t> ok(nchunk, 1);
        dut::parallel_for(nchunk, 1, [&](size_t c) {
            const size_t a = c * grain, b = std::min(n_iv, a + grain);
            bool good = true;
            for (size_t i = a; i < b; ++i)
                good = good && iv[i].state <= 5 && iv[i].end > iv[i].start && (i == 0 || iv[i].state != iv[i - 1].state);
            ok[c] = good;
        });
        if (std::all_of(ok.begin(), ok.end(), [](uint8_t v) { return v != 0; })) n_seq = 1;
    }
    for (size_t i = 0; i < n_seq; ++i) {
        const uint64_t start = iv[i].start, end = iv[i].end;
        const uint32_t state = iv[i].state;
        if (state > 5 || end <= start) return CL_ERR_INVALID;
        if (!p->has_state) {                                  // first position ever, :128-141
            if (state == CL_REF_N) {
                p->cur_contig = contig; p->cur_start = 0; p->cur_end = start + 1; p->cur_state = state;
                p->has_state = true;
            } else {
                if (start > 0) {
                    p->cur_contig = contig; p->cur_start = 0; p->cur_end = start; p->cur_state = CL_REF_N;
                    p->has_state = true;
                    p->write_state();
                }
                p->cur_contig = contig; p->cur_start = start; p->cur_end = start + 1; p->cur_state = state;
                p->has_state = true;
            }
            p->cur_end = end;                                 // the remaining positions of the run extend it
            continue;
        }
        if (p->cur_contig == contig && p->cur_state == state) {
            p->cur_end = end;                                 // :144-146
        } else {
            p->write_state();                                 // :147-151
            p->cur_contig = contig; p->cur_start = start; p->cur_end = end; p->cur_state = state;
        }
    }
    if (n_seq < n_iv) {
        // intervals 1 .. n-1: the current state (interval 0, as the rules above left it) is written, then every
        // interval but the last; the last becomes the current state
        p->write_state();
        p->flush_text();
        const size_t m = n_iv - 1 - n_seq;                    // lines of intervals n_seq .. n_iv - 2
        const size_t grain = 16384, nchunk = (m + grain - 1) / grain;
        const size_t name_len = strlen(contig);
        std::vector<std::string> part(nchunk);
        dut::parallel_for(nchunk, 1, [&](size_t c) {
            const size_t a = n_seq + c * grain, b = std::min(n_seq + m, a + grain);
            std::string &o = part[c];
            o.reserve((b - a) * (name_len + 48));
            for (size_t i = a; i < b; ++i) {
                o.append(contig, name_len); o.push_back('\t');
                dut_profiler::put_u64(o, iv[i].start); o.push_back('\t');
                dut_profiler::put_u64(o, iv[i].end); o.push_back('\t');
                o += kStateNames[iv[i].state]; o.push_back('\n');
            }
        });
        for (const std::string &o : part) fwrite(o.data(), 1, o.size(), p->bed);
        const cl_interval &last = iv[n_iv - 1];
        p->cur_contig = contig; p->cur_start = last.start; p->cur_end = last.end; p->cur_state = last.state;
    }
    p->write_state();                                         // finish_contig, :64-66 (state kept)
    return CL_OK;
}

// DUT_TIMING=1: wall-clock of the host stages on stderr (tooling; off by default)
static double dut_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static bool dut_timing_on() { static const bool on = getenv("DUT_TIMING") && *getenv("DUT_TIMING") == '1'; return on; }
static void dut_stage_time(const char *what, double &t0)
{
    if (!dut_timing_on()) return;
    const double t1 = dut_now();
    fprintf(stderr, "[dut-timing] %-28s %8.1f ms\n", what, (t1 - t0) * 1e3);
    t0 = t1;
}

// 64-bit hashes of every record's name (not value-initialised: first touched by the threads that fill it)
static dut::Scratch<uint64_t> hash_all_names(const dut_records *rec)
{
    dut::Scratch<uint64_t> h_buf(rec->n ? rec->n : 1);
    if (!rec->qname_off) return h_buf;
    uint64_t *h = h_buf.get();
    NameSet hasher; hasher.rec = rec;
    dut::parallel_for(rec->n, 65536, [&](size_t i) { h[i] = hasher.hash_of(i); });
    return h_buf;
}

// distinct read names among the accepted reads (contig_profiler.rs:54-57: the HashSet of qnames), exactly.
// h: the names' hashes (hash_all_names), which do not depend on the admission and can be had beside it.
//
// The accepted reads are partitioned by the top bits of their hash -- per-chunk histograms, offsets, scatter, all
// parallel -- and the scatter moves the names themselves: every class gets its hashes and its names as contiguous
// arrays (a few thousand names, some tens of kilobytes), so that the exact set of a class (open addressing on the
// hash, full comparison of the bytes on a hash match) works in cache instead of reaching into the contig's name
// array at random twice per insert.  Classes are counted in parallel and the counts add up.
static uint32_t count_unique_names(const dut_records *rec, const uint8_t *accepted, uint64_t nacc, const uint64_t *h)
{
    double tm = dut_now();
    if (!rec->qname_off || !nacc) return 0;
    const int bits = nacc > (1u << 22) ? 12 : (nacc > (1u << 16) ? 8 : 0);
    const size_t kClasses = (size_t)1 << bits;
    const size_t grain = 1u << 18, nchunk = (rec->n + grain - 1) / grain;
    auto cls = [&](uint64_t hv) -> size_t { return bits ? (size_t)(hv >> (64 - bits)) : 0; };
    // reads and name bytes per (chunk, class)
    std::vector<uint32_t> cnt(nchunk * kClasses, 0);
    std::vector<uint64_t> byt(nchunk * kClasses, 0);
    dut::parallel_for(nchunk, 1, [&](size_t c) {
        uint32_t *cc = cnt.data() + c * kClasses;
        uint64_t *bc = byt.data() + c * kClasses;
        const size_t b = std::min<size_t>(rec->n, (c + 1) * grain);
        for (size_t i = c * grain; i < b; ++i)
            if (accepted[i]) { const size_t k = cls(h[i]); cc[k] += 1; bc[k] += rec->qname_off[i + 1] - rec->qname_off[i]; }
    });
    // class-major, chunk-minor exclusive prefixes: where chunk c writes its reads / bytes of class k
    std::vector<uint64_t> rstart(kClasses + 1, 0), bstart(kClasses + 1, 0);
    std::vector<uint64_t> roff(nchunk * kClasses), boff(nchunk * kClasses);
    {
        uint64_t rr = 0, bb = 0;
        for (size_t k = 0; k < kClasses; ++k) {
            rstart[k] = rr; bstart[k] = bb;
            for (size_t c = 0; c < nchunk; ++c) { roff[c * kClasses + k] = rr; boff[c * kClasses + k] = bb; rr += cnt[c * kClasses + k]; bb += byt[c * kClasses + k]; }
        }
        rstart[kClasses] = rr; bstart[kClasses] = bb;
    }
    const uint64_t total_bytes = bstart[kClasses];
    dut::Scratch<uint64_t> ch_buf(nacc);                                  // hashes, class by class
    dut::Scratch<uint64_t> cn_buf(nacc + 1);                              // offsets of the names in cbytes (global)
    dut::Scratch<uint8_t> cb_buf(total_bytes ? total_bytes : 1);
    uint64_t *ch = ch_buf.get(), *cn = cn_buf.get();
    uint8_t *cb = cb_buf.get();
    dut::parallel_for(nchunk, 1, [&](size_t c) {
        uint64_t *ro = roff.data() + c * kClasses, *bo = boff.data() + c * kClasses;
        const size_t b = std::min<size_t>(rec->n, (c + 1) * grain);
        for (size_t i = c * grain; i < b; ++i) {
            if (!accepted[i]) continue;
            const size_t k = cls(h[i]);
            const uint32_t nl = rec->qname_off[i + 1] - rec->qname_off[i];
            ch[ro[k]] = h[i]; cn[ro[k]] = bo[k];
            memcpy(cb + bo[k], rec->qname + rec->qname_off[i], nl);
            ro[k] += 1; bo[k] += nl;
        }
    });
    dut_stage_time("  names: partition", tm);
    std::vector<uint64_t> per(kClasses, 0);
    dut::parallel_for(kClasses, 16, [&](size_t k) {
        const uint64_t a = rstart[k], b = rstart[k + 1];
        if (a == b) return;
        const uint64_t bend = bstart[k + 1];
        size_t cap = 1024;
        while (cap < (b - a) * 2 + 2) cap *= 2;
        std::vector<uint64_t> th(cap, 0);                                  // hash per slot (0 = empty; hashes are never 0)
        std::vector<uint32_t> ti(cap, 0);                                  // class-local index of the slot's name
        uint64_t distinct = 0;
        for (uint64_t q = a; q < b; ++q) {
            const uint64_t hv = ch[q];
            const uint64_t o = cn[q], ol = (q + 1 < b ? cn[q + 1] : bend) - o;
            size_t j = hv & (cap - 1);
            bool found = false;
            while (th[j]) {
                if (th[j] == hv) {
                    const uint64_t p = a + ti[j];
                    const uint64_t po = cn[p], pl = (p + 1 < b ? cn[p + 1] : bend) - po;
                    if (pl == ol && memcmp(cb + po, cb + o, ol) == 0) { found = true; break; }
                }
                j = (j + 1) & (cap - 1);
            }
            if (!found) { th[j] = hv; ti[j] = (uint32_t)(q - a); ++distinct; }
        }
        per[k] = distinct;
    });
    uint64_t total = 0;
    for (uint64_t v : per) total += v;
    dut_stage_time("  names: sets", tm);
    return (uint32_t)total;
}

static int dut_admit_reads_impl(const cl_options *opt, int32_t tid, uint32_t contig_len, const dut_records *rec,
                    uint8_t *accepted, uint32_t *n_unique_names, uint64_t *n_accepted)
{
    if (!opt || !rec || (!accepted && rec->n)) return CL_ERR_INVALID;
    if (rec->n >= (1ull << 32)) return CL_ERR_RANGE;          // read indices are kept in 32 bits below (the engine stops at 2^29 reads)
    const uint64_t maxcnt = opt->max_depth > 0 ? opt->max_depth : 500;   // mod.rs:56-60
    // Ends of the reads the pileup list holds.  When the cursor sits on start position s the
    // list holds exactly the appended reads with end >= s (reads that ended earlier were freed
    // while the columns before s were produced; a read ending AT s is freed only when column s
    // itself is walked).
    // `live`: the ends, kept ascending in a ring (reads arrive in start order and their ends are nearly
    // sorted, so a new end is placed by a short insertion from the back; expired ends leave at the front).
    std::vector<uint64_t> ring(1024);
    size_t head = 0, cnt = 0;                                  // ring[(head + i) & mask], i < cnt, ascending
    auto live_push = [&](uint64_t e) {
        if (cnt == ring.size()) {                              // grow, unrolled to index 0
            std::vector<uint64_t> nr(ring.size() * 2);
            for (size_t i = 0; i < cnt; ++i) nr[i] = ring[(head + i) & (ring.size() - 1)];
            ring.swap(nr); head = 0;
        }
        const size_t mask = ring.size() - 1;
        size_t i = cnt;
        while (i > 0 && ring[(head + i - 1) & mask] > e) { ring[(head + i) & mask] = ring[(head + i - 1) & mask]; --i; }
        ring[(head + i) & mask] = e;
        ++cnt;
    };
    // reference spans of all reads up front, in parallel (the sequential rule below only compares numbers)
    double tm = dut_now();
    // (not value-initialised: the pages are first touched by the threads that fill them)
    dut::Scratch<uint64_t> rlen_buf(rec->n ? rec->n : 1);
    uint64_t *rlen = rlen_buf.get();
    dut::parallel_for(rec->n, dut::grain_for(rec->n, 65536), [&](size_t i) { rlen[i] = ref_length(rec->cigar + rec->cigar_off[i], rec->cigar_off[i + 1] - rec->cigar_off[i]); });
    dut_stage_time("  admit: spans", tm);
    bool any_pushed = false;
    int64_t cur_start = -1;
    uint64_t nacc = 0;
    // Fast path.  The list never holds more reads than have started within the longest span before the
    // cursor: when the records are sorted and fewer than maxcnt records start within max_span of any
    // record, the cap cannot bite, and what is left of the rule is independent per read: kept iff it
    // is yielded by the fetch, mapped, and spans reference positions (a spanless read is appended or not
    // depending on the branch, but never counted).  Checked and applied in parallel.
    bool fast = rec->n > 0;
    if (fast) {
        const size_t grain = 65536, nchunk = (rec->n + grain - 1) / grain;
        std::vector<uint64_t> c_maxrl(nchunk, 0);
        std::vector<uint8_t> c_sorted(nchunk, 1);
        dut::parallel_for(nchunk, 1, [&](size_t c) {
            const size_t a = c * grain, b = std::min<size_t>(rec->n, a + grain);
            uint64_t m = 0; bool ok = true;
            for (size_t i = a; i < b; ++i) {
                m = std::max(m, rlen[i]);
                if (i + 1 < rec->n && rec->pos[i + 1] < rec->pos[i]) ok = false;
                if (rec->pos[i] < 0) ok = false;
            }
            c_maxrl[c] = m; c_sorted[c] = ok;
        });
        uint64_t max_rl = 0;
        for (size_t c = 0; c < nchunk; ++c) { max_rl = std::max(max_rl, c_maxrl[c]); fast = fast && c_sorted[c]; }
        if (fast) {
            std::vector<uint8_t> c_ok(nchunk, 1);
            dut::parallel_for(nchunk, 1, [&](size_t c) {
                const size_t a = c * grain, b = std::min<size_t>(rec->n, a + grain);
                // j: first record that starts at or after pos[i] - max_rl
                const int64_t key0 = (int64_t)rec->pos[a] - (int64_t)max_rl;
                size_t j = (size_t)(std::lower_bound(rec->pos, rec->pos + a, key0, [](int32_t v, int64_t k) { return (int64_t)v < k; }) - rec->pos);
                bool ok = true;
                for (size_t i = a; i < b && ok; ++i) {
                    const int64_t key = (int64_t)rec->pos[i] - (int64_t)max_rl;
                    while ((int64_t)rec->pos[j] < key) ++j;
                    if (i - j + 1 >= maxcnt) ok = false;          // i - j records before this one could still be listed
                }
                c_ok[c] = ok;
            });
            for (size_t c = 0; c < nchunk; ++c) fast = fast && c_ok[c];
        }
        if (fast) {
            std::vector<uint64_t> c_acc(nchunk, 0);
            dut::parallel_for(nchunk, 1, [&](size_t c) {
                const size_t a = c * grain, b = std::min<size_t>(rec->n, a + grain);
                uint64_t k = 0;
                for (size_t i = a; i < b; ++i) {
                    const bool keep = (int64_t)rec->pos[i] < (int64_t)contig_len && !(rec->flag[i] & 0x4) && rlen[i] > 0;
                    accepted[i] = keep ? 1 : 0;
                    k += keep;
                }
                c_acc[c] = k;
            });
            for (size_t c = 0; c < nchunk; ++c) nacc += c_acc[c];
        }
    }
    for (uint64_t i = 0; !fast && i < rec->n; ++i) {
        accepted[i] = 0;
        const int64_t p = rec->pos[i];
        if (p >= (int64_t)contig_len) continue;              // not yielded by fetch((tid,0,len))
        if (rec->flag[i] & 0x4) continue;                    // BAM_FUNMAP: bam_plp_push skips it
        const uint64_t rl = rlen[i];
        const uint64_t end = (uint64_t)p + rl;
        bool appended;
        if (!any_pushed) {
            // the iterator starts at (tid 0, pos 0): the cap test can only see an empty list here
            appended = end > 0 || tid > 0;
            any_pushed = true; cur_start = p;
        } else if (p == cur_start) {
            // not the first read at this start: the cursor sits here, the cap applies
            if (cnt >= maxcnt) continue;
            appended = end > (uint64_t)p;
        } else {
            if (p < cur_start) return CL_ERR_UNSORTED;
            cur_start = p;
            const size_t mask = ring.size() - 1;
            while (cnt && ring[head & mask] < (uint64_t)p) { head = (head + 1) & mask; --cnt; }
            appended = true;                                  // cursor still on the previous start
        }
        if (!appended) continue;
        live_push(end);
        if (rl > 0) { accepted[i] = 1; ++nacc; }
    }
    dut_stage_time("  admit: cap rule", tm);
    if (n_accepted) *n_accepted = nacc;
    if (n_unique_names) {
        const dut::Scratch<uint64_t> h = hash_all_names(rec);
        *n_unique_names = count_unique_names(rec, accepted, nacc, h.get());
    }
    return CL_OK;
}

int dut_admit_reads(const cl_options *opt, int32_t tid, uint32_t contig_len, const dut_records *rec,
                    uint8_t *accepted, uint32_t *n_unique_names, uint64_t *n_accepted)
{
    // no exception leaves the library through the C ABI
    try { return dut_admit_reads_impl(opt, tid, contig_len, rec, accepted, n_unique_names, n_accepted); }
    catch (const std::bad_alloc &) { return CL_ERR_NOMEM; }
    catch (...) { return CL_ERR_INVALID; }
}



int dut_process_single_contig(cl_ctx *ctx, dut_profiler *prof, dut_contig_stats *stats, const cl_options *opt,
                              const char *contig_name, int32_t tid, uint32_t contig_len, const uint8_t *ref,
                              uint64_t ref_len, const dut_records *rec)
{
    if (!ctx || !prof || !stats || !opt || !contig_name || !rec) return CL_ERR_INVALID;
    uint64_t counts[6]; const cl_interval *iv = nullptr; size_t niv = 0;
    int rc = dut_process_single_contig_runs(ctx, stats, opt, tid, contig_len, ref, ref_len, rec, counts, &iv, &niv);
    if (rc != CL_OK) return rc;
    double tm = dut_now();
    rc = dut_profiler_feed_contig(prof, contig_name, iv, niv, counts);
    if (rc == CL_OK && dut_profiler_finish_plot(prof, contig_name, contig_len) < 0) rc = CL_ERR_INVALID;   // finish_contig, :67-84
    dut_stage_time("BED lines", tm);
    return rc;
}

static int dut_process_single_contig_runs_impl(cl_ctx *ctx, dut_contig_stats *stats, const cl_options *opt, int32_t tid,
                                   uint32_t contig_len, const uint8_t *ref, uint64_t ref_len, const dut_records *rec,
                                   uint64_t state_counts[6], const cl_interval **intervals, size_t *n_intervals)
{
    if (!ctx || !stats || !opt || !rec || !state_counts || !intervals || !n_intervals) return CL_ERR_INVALID;
    double tm = dut_now();
    int rc = cl_contig_begin(ctx, tid, contig_len, ref, ref_len);
    if (rc != CL_OK) return rc;
    // Every early return below abandons the contig: a quality prefetch in flight still reads `rec->qual` and holds the
    // device's staging ring, and the caller frees the records (and may destroy the context) right after an error.
    // Declared before the helper threads so that it runs after they are joined.
    struct AbortGuard { cl_ctx *c; bool armed = true; ~AbortGuard() { if (armed) (void)cl_contig_abort(c); } } abandon{ctx};
    // Reads at or past contig_len (never yielded by the region fetch) are the sorted tail and are cut off.  The quality
    // bytes of the rest start towards the device now, beside the admission below (the tile is pushed as the records lie;
    // should it turn out that it cannot be -- unsorted input, leading reads to drop -- the engine discards the prefetch).
    uint64_t a0 = 0, n_keep = rec->n;
    while (n_keep > 0 && (int64_t)rec->pos[n_keep - 1] >= (int64_t)contig_len) --n_keep;
    // what is about to be pushed, so that the engine sizes its staging arrays once and has its device buffers allocated
    // while the reads are admitted (a hint: never required)
    if (n_keep > 0) {
        rc = cl_contig_reserve(ctx, n_keep, rec->cigar_off[n_keep], rec->qual_off[n_keep] - rec->qual_off[0]);
        if (rc != CL_OK) return rc;
    }
    if (n_keep > 0 && rec->pos[0] >= 0 && rec->qual) {
        rc = cl_contig_prefetch_qual(ctx, rec->qual + rec->qual_off[0], rec->qual_off[n_keep] - rec->qual_off[0]);
        if (rc != CL_OK) return rc;
    }
    if (!rec->qual && !rec->pass_bits && rec->n && rec->qual_off[rec->n] > rec->qual_off[0]) return CL_ERR_INVALID;   // neither bytes nor bits
    std::vector<uint8_t> acc(rec->n ? rec->n : 1);
    uint32_t n_names = 0; uint64_t n_acc = 0;
    // the names' hashes do not depend on the admission: beside it, on their own thread
    dut::Scratch<uint64_t> name_hash;
    bool hashed = false;
    dut::Thread hasher = dut::spawn_or_run([&]() { name_hash = hash_all_names(rec); hashed = true; });
    rc = dut_admit_reads(opt, tid, contig_len, rec, acc.data(), nullptr, &n_acc);
    if (rc != CL_OK) return rc;
    dut_stage_time("admit", tm);
    // the distinct-name count is needed at the very end only: on its own thread beside the push, the kernels and the
    // read-back (joined before this function returns, also on every error path: dut::Thread joins in its destructor)
    bool names_done = false;
    dut::Thread names = dut::spawn_or_run([&]() {
        if (hasher.joinable()) hasher.join();
        if (!hashed) return;
        n_names = count_unique_names(rec, acc.data(), n_acc, name_hash.get());
        names_done = true;
    });
    // One tile, no copy of the quality bytes: the records of the contig are pushed as they lie in
    // `rec`.  Reads the pileup would not hold (FUNMAP, the depth cap, reads without a reference span)
    // stay in the tile with their CIGAR operations rewritten to hard clips, which consume neither
    // reference nor query: they then touch no position and add to no sum.  Reads at or past
    // contig_len were cut off above, so are unaccepted reads in front of position 0.
    while (a0 < n_keep && rec->pos[a0] < 0 && !acc[a0]) ++a0;
    bool in_order = true;
    uint64_t n_in = 0;
    {   // sortedness and the number of accepted reads in [a0, n_keep), in chunks on all host threads
        const size_t grain = 1u << 18, nchunk = n_keep > a0 ? (n_keep - a0 + grain - 1) / grain : 0;
        std::vector<uint64_t> c_in(nchunk, 0);
        std::vector<uint8_t> c_ok(nchunk, 1);
        dut::parallel_for(nchunk, 1, [&](size_t c) {
            const uint64_t b0 = a0 + c * grain, b1 = std::min<uint64_t>(n_keep, b0 + grain);
            uint64_t k = 0; bool ok = true;
            for (uint64_t i = b0; i < b1; ++i) { k += acc[i]; if (i > a0 && rec->pos[i] < rec->pos[i - 1]) ok = false; }
            c_in[c] = k; c_ok[c] = ok;
        });
        for (size_t c = 0; c < nchunk; ++c) { n_in += c_in[c]; in_order = in_order && c_ok[c]; }
    }
    dut_stage_time("  push: order check", tm);
    if (in_order && a0 < n_keep) {
        // the patched copy lives in a per-thread scratch buffer that is kept from contig to contig (a fresh,
        // zero-filled vector of a contig's 40 MB of CIGAR words cost more than the patching)
        struct Scratch { uint32_t *p = nullptr; size_t cap = 0; ~Scratch() { free(p); } };
        static thread_local Scratch patched;
        const uint32_t *cig = rec->cigar;
        if (n_in != n_keep - a0) {
            const size_t need = rec->cigar_off[n_keep];
            if (need > patched.cap) {
                free(patched.p); patched.p = nullptr; patched.cap = 0;
                patched.p = static_cast<uint32_t *>(malloc((need + need / 8 + 16) * sizeof(uint32_t)));
                if (!patched.p) return CL_ERR_NOMEM;
                patched.cap = need + need / 8 + 16;
            }
            uint32_t *pp = patched.p;
            dut::parallel_for(n_keep, dut::grain_for(n_keep, 1u << 16), [&](size_t i) {           // reads in front of a0 keep their words: they are not in the tile
                const bool keep = i < a0 || acc[i];
                for (uint32_t k = rec->cigar_off[i]; k < rec->cigar_off[i + 1]; ++k) pp[k] = keep ? rec->cigar[k] : ((rec->cigar[k] & ~15u) | 5u);
            });
            cig = patched.p;
        }
        dut_stage_time("  push: patched CIGARs", tm);
        if (rec->pass_bits) {
            // the packed variant: the reader has taken the base-quality test (dut_bam_read_contig_bits)
            cl_read_tile_bits t;
            t.n_reads = n_keep - a0; t.pos = rec->pos + a0; t.mapq = rec->mapq + a0; t.cigar_off = rec->cigar_off + a0;
            t.cigar = cig; t.qual_off = rec->qual_off + a0; t.pass_bits = rec->pass_bits; t.pass_sum = rec->pass_sum + a0;
            rc = cl_push_reads_bits(ctx, &t);
        } else {
            cl_read_tile t;
            t.n_reads = n_keep - a0; t.pos = rec->pos + a0; t.mapq = rec->mapq + a0; t.cigar_off = rec->cigar_off + a0;
            t.cigar = cig; t.qual_off = rec->qual_off + a0; t.qual = rec->qual;
            rc = cl_push_reads(ctx, &t);
        }
        if (rc != CL_OK) return rc;
    } else if (a0 < n_keep) {
        // records whose skipped reads are out of order: copy the accepted ones out, tile by tile
        const uint64_t kTile = 1u << 20;
        std::vector<int32_t> pos; std::vector<uint8_t> mapq; std::vector<uint32_t> coff, cig;
        std::vector<uint64_t> qoff; std::vector<uint8_t> qual;
        std::vector<uint64_t> pbits; std::vector<uint32_t> psum;       // the packed variant's bits and sums of the copied reads
        uint64_t i = 0;
        while (i < rec->n) {
            pos.clear(); mapq.clear(); cig.clear(); qual.clear(); pbits.assign(1, 0ull); psum.clear();
            coff.assign(1, 0u); qoff.assign(1, 0ull);
            uint64_t nb = 0;
            for (; i < rec->n && pos.size() < kTile; ++i) {
                if (!acc[i]) continue;
                pos.push_back(rec->pos[i]); mapq.push_back(rec->mapq[i]);
                cig.insert(cig.end(), rec->cigar + rec->cigar_off[i], rec->cigar + rec->cigar_off[i + 1]);
                const uint64_t ql = rec->qual_off[i + 1] - rec->qual_off[i];
                if (rec->pass_bits) {
                    pbits.resize(((nb + ql + 63) >> 6) + 1, 0ull);
                    for (uint64_t k = 0; k < ql; ++k) {
                        const uint64_t sb = rec->qual_off[i] + k;
                        if ((rec->pass_bits[sb >> 6] >> (sb & 63ull)) & 1ull) pbits[(nb + k) >> 6] |= 1ull << ((nb + k) & 63ull);
                    }
                    psum.push_back(rec->pass_sum[i]);
                } else qual.insert(qual.end(), rec->qual + rec->qual_off[i], rec->qual + rec->qual_off[i + 1]);
                nb += ql;
                coff.push_back((uint32_t)cig.size()); qoff.push_back(nb);
            }
            if (pos.empty()) continue;
            if (rec->pass_bits) {
                cl_read_tile_bits t;
                t.n_reads = pos.size(); t.pos = pos.data(); t.mapq = mapq.data(); t.cigar_off = coff.data();
                t.cigar = cig.data(); t.qual_off = qoff.data(); t.pass_bits = pbits.data(); t.pass_sum = psum.data();
                rc = cl_push_reads_bits(ctx, &t);
            } else {
                cl_read_tile t;
                t.n_reads = pos.size(); t.pos = pos.data(); t.mapq = mapq.data(); t.cigar_off = coff.data();
                t.cigar = cig.data(); t.qual_off = qoff.data(); t.qual = qual.data();
                rc = cl_push_reads(ctx, &t);
            }
            if (rc != CL_OK) return rc;
        }
    }
    dut_stage_time("compact + push", tm);
    cl_contig_summary sum; const cl_interval *iv = nullptr; size_t niv = 0;
    rc = cl_contig_finish(ctx, &sum, &iv, &niv);
    if (rc != CL_OK) return rc;
    dut_stage_time("upload + kernels + collect", tm);
    if (names.joinable()) names.join();
    if (!names_done) return CL_ERR_NOMEM;
    dut_stage_time("wait for the name count", tm);
    for (int i = 0; i < 6; ++i) state_counts[i] = sum.state_counts[i];
    *intervals = iv; *n_intervals = niv;
    stats->length = contig_len;
    stats->n_covered_bases = sum.n_covered_bases;
    stats->summed_coverage = sum.summed_coverage;
    stats->summed_baseq = sum.summed_baseq;
    stats->summed_mapq = sum.summed_mapq;
    stats->quality_bases = sum.quality_bases;
    stats->n_reads = n_names;
    stats->reserved = 0;
    abandon.armed = false;
    return CL_OK;
}

int dut_process_single_contig_runs(cl_ctx *ctx, dut_contig_stats *stats, const cl_options *opt, int32_t tid,
                                   uint32_t contig_len, const uint8_t *ref, uint64_t ref_len, const dut_records *rec,
                                   uint64_t state_counts[6], const cl_interval **intervals, size_t *n_intervals)
{
    // no exception leaves the library through the C ABI
    try { return dut_process_single_contig_runs_impl(ctx, stats, opt, tid, contig_len, ref, ref_len, rec, state_counts, intervals, n_intervals); }
    catch (const std::bad_alloc &) { return CL_ERR_NOMEM; }
    catch (...) { return CL_ERR_INVALID; }
}


void dut_contig_derive(const dut_contig_stats *s, dut_contig_derived *o)
{
    o->coverage_percent = s->length > 0 ? ((double)s->n_covered_bases / (double)s->length) * 100.0 : 0.0;
    o->average_depth = s->n_covered_bases > 0 ? (double)s->summed_coverage / (double)s->n_covered_bases : 0.0;
    o->average_mapq = s->quality_bases > 0 ? (double)s->summed_mapq / (double)s->quality_bases : 0.0;
    o->average_baseq = s->quality_bases > 0 ? (double)s->summed_baseq / (double)s->quality_bases : 0.0;
    if (s->quality_bases > 0) {
        if (o->average_baseq >= 30.0) o->q30_percentage = 100.0;
        else if (o->average_baseq < 20.0) o->q30_percentage = 0.0;
        else o->q30_percentage = ((o->average_baseq - 20.0) / 10.0) * 100.0;
    } else o->q30_percentage = 0.0;
}

int dut_compare_contig_names(const char *a, const char *b)
{
    // split at the first ASCII digit or 'X' 'Y' 'M' (report.rs:385-393)
    auto split = [](const std::string &s) {
        size_t i = 0;
        for (; i < s.size(); ++i) { char c = s[i]; if ((c >= '0' && c <= '9') || c == 'X' || c == 'Y' || c == 'M') break; }
        return i;
    };
    // (category, number): numeric first, then X, Y, M/MT, then the rest (report.rs:355-369)
    auto order = [](const std::string &s, int &cat, uint32_t &num) {
        size_t st = (!s.empty() && s[0] == '+') ? 1 : 0;     // u32::from_str accepts a leading '+'
        bool ok = s.size() > st; uint64_t v = 0;
        for (size_t i = st; ok && i < s.size(); ++i) {
            if (s[i] < '0' || s[i] > '9') ok = false;
            else { v = v * 10 + (uint64_t)(s[i] - '0'); if (v > 0xFFFFFFFFull) ok = false; }
        }
        num = 0;
        if (ok) { cat = 0; num = (uint32_t)v; }
        else if (s == "X") cat = 1;
        else if (s == "Y") cat = 2;
        else if (s == "M" || s == "MT") cat = 3;
        else cat = 4;
    };
    const std::string sa(a), sb(b);
    const size_t ia = split(sa), ib = split(sb);
    const int pc = sa.compare(0, ia, sb, 0, ib);
    if (pc != 0) return pc < 0 ? -1 : 1;
    const std::string xa = sa.substr(ia), xb = sb.substr(ib);
    int ca, cb; uint32_t na, nb;
    order(xa, ca, na); order(xb, cb, nb);
    if (ca != cb) return ca < cb ? -1 : 1;
    if (ca == 0) return na < nb ? -1 : (na > nb ? 1 : 0);
    const int c = xa.compare(xb);
    return c < 0 ? -1 : (c > 0 ? 1 : 0);
}

void dut_genome_summary_build(const dut_contig_stats *stats, const uint64_t *callable, size_t n, dut_genome_summary *out)
{
    uint64_t total_bases = 0, callable_bases = 0, q30_bases = 0, total_quality_positions = 0, total_unique_reads = 0;
    double total_depth = 0.0, total_mapq = 0.0, total_baseq = 0.0;
    for (size_t i = 0; i < n; ++i) {
        dut_contig_derived d;
        dut_contig_derive(&stats[i], &d);
        const double len = (double)stats[i].length;
        total_bases += stats[i].length;
        callable_bases += callable[i];
        total_depth += d.average_depth * len;
        total_mapq += d.average_mapq * len;
        total_baseq += d.average_baseq * len;
        const double q = d.q30_percentage / 100.0 * len;     // `as u64`: saturating, toward zero
        q30_bases += q <= 0.0 ? 0ull : (q >= 18446744073709551615.0 ? ~0ull : (uint64_t)q);
        total_quality_positions += stats[i].length;
        total_unique_reads += stats[i].n_reads;
    }
    out->total_bases = total_bases;
    out->callable_bases = callable_bases;
    out->callable_percentage = total_bases > 0 ? ((double)callable_bases / (double)total_bases) * 100.0 : 0.0;
    out->average_depth = total_bases > 0 ? total_depth / (double)total_bases : 0.0;
    out->average_mapq = total_quality_positions > 0 ? total_mapq / (double)total_quality_positions : 0.0;
    out->average_baseq = total_quality_positions > 0 ? total_baseq / (double)total_quality_positions : 0.0;
    out->q30_percentage = total_quality_positions > 0 ? ((double)q30_bases / (double)total_quality_positions) * 100.0 : 0.0;
    out->total_unique_reads = total_unique_reads;
    out->contigs_analyzed = n;
}

} // extern "C"
