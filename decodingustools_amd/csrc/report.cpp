// report.cpp -- implementation of include/dut_report.h: BamStats, platform inference and the
// summary.json text of the `coverage` command.  Host-only; nothing here touches the device.
#include "../../include/dut_report.h"
#include "../../include/dut_bam.h"

#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <string_view>
#include <vector>

namespace {

using sv = std::string_view;

bool contains(sv s, sv needle) { return s.find(needle) != sv::npos; }
bool starts_with(sv s, sv p) { return s.size() >= p.size() && s.compare(0, p.size(), p) == 0; }
size_t count_char(sv s, char c) { return (size_t)std::count(s.begin(), s.end(), c); }
std::vector<sv> split(sv s, char c)
{
    std::vector<sv> out;
    size_t a = 0;
    for (;;) {
        const size_t b = s.find(c, a);
        if (b == sv::npos) { out.push_back(s.substr(a)); break; }
        out.push_back(s.substr(a, b - a));
        a = b + 1;
    }
    return out;
}
bool is_hex(sv s)
{
    for (char ch : s) if (!((ch >= '0' && ch <= '9') || (ch >= 'a' && ch <= 'f') || (ch >= 'A' && ch <= 'F'))) return false;
    return true;
}
std::string ascii_lower(sv s) { std::string o(s); for (char &c : o) if (c >= 'A' && c <= 'Z') c = (char)(c + 32); return o; }
std::string ascii_upper(sv s) { std::string o(s); for (char &c : o) if (c >= 'a' && c <= 'z') c = (char)(c - 32); return o; }

// std::str::from_utf8(record.qname()).is_ok()  (bam_stats.rs:81)
bool valid_utf8(const uint8_t *p, size_t n)
{
    size_t i = 0;
    while (i < n) {
        const uint8_t c = p[i];
        if (c < 0x80) { ++i; continue; }
        size_t need; uint32_t cp;
        if (c >= 0xC2 && c <= 0xDF) { need = 1; cp = c & 0x1Fu; }
        else if (c >= 0xE0 && c <= 0xEF) { need = 2; cp = c & 0x0Fu; }
        else if (c >= 0xF0 && c <= 0xF4) { need = 3; cp = c & 0x07u; }
        else return false;
        if (i + need >= n) return false;                         // truncated sequence
        for (size_t k = 1; k <= need; ++k) { if ((p[i + k] & 0xC0u) != 0x80u) return false; cp = (cp << 6) | (p[i + k] & 0x3Fu); }
        if ((need == 2 && (cp < 0x800u || (cp >= 0xD800u && cp <= 0xDFFFu))) || (need == 3 && (cp < 0x10000u || cp > 0x10FFFFu))) return false;
        i += need + 1;
    }
    return true;
}

int detect_platform(sv q)
{
    // Oxford Nanopore: UUID-like names, or "ch" + "read" in a long name (platform_inference.rs:21-43)
    if (q.size() > 30 && (contains(q, "-") || contains(q, "_"))) {
        const std::vector<sv> parts = split(q, '-');
        if (parts.size() == 5) {
            const bool is_uuid = parts[0].size() == 8 && parts[1].size() == 4 && parts[2].size() == 4 &&
                                 parts[3].size() == 4 && parts[4].size() >= 12;
            bool all_hex = true;
            for (sv p : parts) all_hex = all_hex && is_hex(p);
            if (is_uuid && all_hex) return DUT_PLATFORM_NANOPORE;
        }
        if (contains(q, "ch") && contains(q, "read")) return DUT_PLATFORM_NANOPORE;
    }
    // PacBio: m<instrument>_<date>_<time>/<zmw>/<type>  (:47-52)
    if (starts_with(q, "m") && contains(q, "/")) {
        const std::vector<sv> parts = split(q, '/');
        if (parts.size() >= 2 && contains(parts[0], "_")) return DUT_PLATFORM_PACBIO;
    }
    // MGI (:57-81)
    if (q.size() > 15) {
        const std::string prefix = ascii_upper(q.substr(0, 5));
        if (starts_with(prefix, "V300") || starts_with(prefix, "E100") || starts_with(prefix, "CL100") ||
            starts_with(prefix, "G400") || starts_with(prefix, "G99"))
            return DUT_PLATFORM_MGI;
        if (count_char(q, ':') >= 6) {
            const std::vector<sv> parts = split(q, ':');
            if (starts_with(parts[0], "V") || starts_with(parts[0], "E") || starts_with(parts[0], "CL") || starts_with(parts[0], "G")) {
                if (parts.size() >= 3 && starts_with(parts[2], "L")) return DUT_PLATFORM_MGI;
            }
        }
    }
    // Illumina (:85-87)
    if (count_char(q, ':') >= 6) return DUT_PLATFORM_ILLUMINA;
    return DUT_PLATFORM_UNKNOWN;
}

struct Parsed { bool ok = false; sv instrument, flow_cell; bool has_fc = false; };

Parsed parse_name(int platform, sv q)
{
    Parsed r;
    switch (platform) {
    case DUT_PLATFORM_ILLUMINA: {                            // :95-104
        const std::vector<sv> parts = split(q, ':');
        if (parts.size() >= 3) { r.ok = true; r.instrument = parts[0]; r.flow_cell = parts[2]; r.has_fc = true; }
        break;
    }
    case DUT_PLATFORM_PACBIO: {                              // :110-122
        const size_t slash = q.find('/');
        if (slash != sv::npos) {
            const sv movie = q.substr(0, slash);
            if (starts_with(movie, "m")) {
                const size_t us = movie.find('_');
                if (us != sv::npos) { r.ok = true; r.instrument = movie.substr(0, us); }
            }
        }
        break;
    }
    case DUT_PLATFORM_NANOPORE: {                            // :128-157
        r.ok = true;
        if (q.size() > 30 && contains(q, "-") && split(q, '-').size() >= 5) {
            const sv a = q.substr(0, std::min(q.find('_'), q.size()));
            r.instrument = a.substr(0, std::min(a.find('-'), a.size()));
            break;
        }
        const size_t us = q.find('_');
        if (us != sv::npos) { r.instrument = q.substr(0, us); break; }
        r.instrument = "nanopore";
        break;
    }
    case DUT_PLATFORM_MGI: {                                 // :163-190
        if (count_char(q, ':') >= 3) {
            const std::vector<sv> parts = split(q, ':');
            if (parts.size() >= 3) { r.ok = true; r.instrument = parts[0]; r.flow_cell = parts[1]; r.has_fc = true; break; }
        }
        if (q.size() > 10) {
            const size_t l_pos = q.find('L');
            if (l_pos != sv::npos) {
                const sv rest = q.substr(l_pos);
                if (rest.find('C') != sv::npos) {
                    const size_t end_pos = std::min(rest.find('R'), rest.size());
                    r.ok = true; r.instrument = q.substr(0, l_pos); r.flow_cell = rest.substr(0, end_pos); r.has_fc = true;
                }
            }
        }
        break;
    }
    default: break;
    }
    return r;
}

const char *specific_platform(int platform, const char *top)
{
    const sv id = top ? sv(top) : sv();
    switch (platform) {
    case DUT_PLATFORM_PACBIO:
        if (top) {
            if (starts_with(id, "m84")) return "PacBio Revio";
            if (starts_with(id, "m64")) return "PacBio Sequel II/IIe";
            if (starts_with(id, "m54")) return "PacBio Sequel";
        }
        return "PacBio";
    case DUT_PLATFORM_NANOPORE: return "Oxford Nanopore";
    case DUT_PLATFORM_MGI:
        if (top) {
            if (starts_with(id, "V300")) return "MGI DNBSEQ/MGISEQ-2000";
            if (starts_with(id, "E100")) return "MGI MGISEQ-200";
            if (starts_with(id, "CL100")) return "MGI MGISEQ-T7";
            if (starts_with(id, "G400")) return "MGI DNBSEQ-G400";
            if (starts_with(id, "G99")) return "MGI MGISEQ-T1";
        }
        return "MGI DNBseq";
    case DUT_PLATFORM_ILLUMINA:
        if (top) {
            // first char of the id; an empty id gives ' ' in the reference, i.e. "Unknown Illumina"
            switch (id.empty() ? ' ' : id[0]) {
            case 'A': case 'a': return "NovaSeq";
            case 'D': case 'd': return "HiSeq 2500";
            case 'J': case 'j': return "HiSeq 3000";
            case 'K': case 'k': return "HiSeq 4000";
            case 'E': case 'e': return "HiSeq X";
            case 'N': case 'n': return "NextSeq";
            case 'M': case 'm': return "MiSeq";
            case 'V': case 'v': return "NovaSeq X";
            case 'F': case 'f': return "iSeq";
            default: return "Unknown Illumina";
            }
        }
        return "Unknown Illumina";
    default: return "Unknown";
    }
}

// the most frequent key; ties go to the smallest key (std::map iterates in key order)
template <class K>
bool top_key(const std::map<K, uint64_t> &m, K *out)
{
    bool any = false; uint64_t best = 0;
    for (const auto &kv : m) if (!any || kv.second > best) { any = true; best = kv.second; *out = kv.first; }
    return any;
}

// ---- a minimal writer for serde_json's PrettyFormatter (two-space indent, "key": value) ----
struct Json {
    std::string s;
    std::vector<bool> first;      // per open container: no member written yet
    void indent() { s.append(2 * first.size(), ' '); }
    void sep()
    {
        if (first.empty()) return;
        s += first.back() ? "\n" : ",\n";
        first.back() = false;
        indent();
    }
    void str(sv v)
    {
        s += '"';
        for (unsigned char c : v) {
            switch (c) {
            case '"': s += "\\\""; break;
            case '\\': s += "\\\\"; break;
            case '\b': s += "\\b"; break;
            case '\f': s += "\\f"; break;
            case '\n': s += "\\n"; break;
            case '\r': s += "\\r"; break;
            case '\t': s += "\\t"; break;
            default:
                if (c < 0x20) { char b[8]; snprintf(b, sizeof(b), "\\u%04x", c); s += b; }
                else s += (char)c;
            }
        }
        s += '"';
    }
    void key(sv k) { sep(); str(k); s += ": "; }
    void open(char c) { s += c; first.push_back(true); }
    void close(char c)
    {
        const bool empty = first.back();
        first.pop_back();
        if (!empty) { s += '\n'; indent(); }
        s += c;
    }
    void u64(uint64_t v) { s += std::to_string(v); }
    void f64(double v) { char b[40]; const size_t n = dut_format_f64(v, b); s.append(b, n); }
};

} // namespace

struct dut_bam_stats {
    uint64_t max_samples = 0;
    uint64_t read_count = 0, total_read_length = 0, paired_reads = 0, paired_count = 0;
    int64_t total_insert_size = 0;
    std::map<uint64_t, uint64_t> length_distribution;
    std::map<int64_t, uint64_t> insert_size_distribution;
    std::map<std::string, uint64_t> flow_cells, instruments;
    std::map<int, uint64_t> platform_counts;
    std::string aligner, reference_build;     // empty until a header is set (BamStats::new)
    mutable std::string top_instrument;
};

extern "C" {

const char *dut_detect_aligner(const char *header_text, size_t len)
{
    const std::string h = ascii_lower(sv(header_text ? header_text : "", header_text ? len : 0));
    if (contains(h, "@pg\tid:bwa-mem2")) return "BWA-MEM2";
    if (contains(h, "@pg\tid:bwa")) return "BWA";
    if (contains(h, "@pg\tid:minimap2")) return "minimap2";
    if (contains(h, "@pg\tid:pbmm2")) return "pbmm2";
    if (contains(h, "@pg\tid:bowtie2")) return "Bowtie2";
    if (contains(h, "@pg\tid:star")) return "STAR";
    if (contains(h, "bwa")) return "BWA";
    if (contains(h, "minimap2")) return "minimap2";
    if (contains(h, "bowtie2")) return "Bowtie2";
    if (contains(h, "star")) return "STAR";
    return "Unknown";
}

const char *dut_reference_build(const char *header_text, size_t len)
{
    const sv h(header_text ? header_text : "", header_text ? len : 0);
    if (contains(h, "AS:GRCh38") || contains(h, "GCA_000001405.15")) return "GRCh38";
    if (contains(h, "AS:GRCh37") || contains(h, "GCA_000001405.1")) return "GRCh37";
    if (contains(h, "AS:CHM13") || contains(h, "GCA_009914755.4")) return "T2T-CHM13v2.0";
    if (contains(h, "chm13") || contains(h, "CHM13") || contains(h, "t2t") || contains(h, "T2T")) return "T2T-CHM13v2.0";
    if (contains(h, "SN:chr1") && contains(h, "LN:248387328") && contains(h, "M5:e469247288ceb332aee524caec92bb22")) return "T2T-CHM13v2.0";
    if (contains(h, "SN:chr1") && contains(h, "LN:248956422")) return "GRCh38";
    if (contains(h, "SN:1") && contains(h, "LN:249250621")) return "GRCh37";
    return "Unknown";
}

int dut_detect_platform_from_qname(const uint8_t *qname, size_t len)
{
    return detect_platform(sv((const char *)qname, qname ? len : 0));
}

int dut_parse_read_name(int platform, const uint8_t *qname, size_t len, const uint8_t **instrument,
                        size_t *instrument_len, const uint8_t **flow_cell, size_t *fc_len)
{
    const Parsed p = parse_name(platform, sv((const char *)qname, qname ? len : 0));
    if (!p.ok) return 0;
    if (instrument) *instrument = (const uint8_t *)p.instrument.data();
    if (instrument_len) *instrument_len = p.instrument.size();
    if (flow_cell) *flow_cell = p.has_fc ? (const uint8_t *)p.flow_cell.data() : nullptr;
    if (fc_len) *fc_len = p.has_fc ? p.flow_cell.size() : 0;
    return 1;
}

const char *dut_infer_specific_platform(int platform, const char *top_instrument)
{
    return specific_platform(platform, top_instrument);
}

dut_bam_stats *dut_bam_stats_new(size_t max_samples)
{
    dut_bam_stats *s = new dut_bam_stats();
    s->max_samples = max_samples;
    return s;
}

void dut_bam_stats_free(dut_bam_stats *s) { delete s; }

void dut_bam_stats_set_header(dut_bam_stats *s, const char *header_text, size_t len)
{
    if (!s) return;
    s->aligner = dut_detect_aligner(header_text, len);
    s->reference_build = dut_reference_build(header_text, len);
}

int dut_bam_stats_add(dut_bam_stats *s, uint64_t index, uint16_t flag, uint32_t l_seq,
                      const uint8_t *qname, size_t qname_len, int32_t tlen)
{
    if (!s || index >= s->max_samples) return 0;                     // bam_stats.rs:61-63
    if (!(flag & 0x100) && !(flag & 0x800)) {                         // primary alignments only (:68)
        s->length_distribution[l_seq] += 1;
        s->read_count += 1;
        s->total_read_length += l_seq;
        if (valid_utf8(qname, qname_len)) {
            const sv q((const char *)qname, qname_len);
            const int platform = detect_platform(q);
            s->platform_counts[platform] += 1;
            const Parsed p = parse_name(platform, q);
            if (p.ok) {
                s->instruments[std::string(p.instrument)] += 1;
                if (p.has_fc) s->flow_cells[std::string(p.flow_cell)] += 1;
            }
        }
        if (flag & 0x1) {                                              // is_paired (:119)
            s->paired_reads += 1;
            if ((flag & 0x2) && (flag & 0x40)) {                       // proper pair, first in template
                const int64_t insert = std::llabs((int64_t)tlen);
                if (insert > 0) {
                    s->insert_size_distribution[insert] += 1;
                    s->total_insert_size += insert;
                    s->paired_count += 1;
                }
            }
        }
    }
    return 1;
}

static int stats_cb(void *ud, uint64_t idx, uint16_t flag, uint32_t l_seq, const uint8_t *qname, size_t qlen, int32_t tlen)
{
    return dut_bam_stats_add((dut_bam_stats *)ud, idx, flag, l_seq, qname, qlen, tlen);
}

int dut_bam_stats_collect(dut_bam_stats *s, const char *bam_path, char *err, size_t err_len)
{
    if (!s || !bam_path) return CL_ERR_INVALID;
    dut_bam *b = dut_bam_open(bam_path, err, err_len);
    if (!b) return CL_ERR_INVALID;
    size_t tl = 0;
    const char *t = dut_bam_header_text(b, &tl);
    dut_bam_stats_set_header(s, t, tl);
    const int rc = dut_bam_sample(b, stats_cb, s);
    if (rc != CL_OK && err && err_len) snprintf(err, err_len, "%s", dut_bam_error(b));
    dut_bam_close(b);
    return rc;
}

const char *dut_bam_stats_aligner(const dut_bam_stats *s) { return s ? s->aligner.c_str() : ""; }
const char *dut_bam_stats_reference_build(const dut_bam_stats *s) { return s ? s->reference_build.c_str() : ""; }
uint64_t dut_bam_stats_read_count(const dut_bam_stats *s) { return s ? s->read_count : 0; }
uint64_t dut_bam_stats_average_read_length(const dut_bam_stats *s) { return (s && s->read_count) ? s->total_read_length / s->read_count : 0; }
uint64_t dut_bam_stats_modal_read_length(const dut_bam_stats *s)
{
    uint64_t k = 0;
    return (s && s->read_count && top_key(s->length_distribution, &k)) ? k : 0;
}
int dut_bam_stats_primary_platform(const dut_bam_stats *s)
{
    int p = DUT_PLATFORM_UNKNOWN;
    if (s) top_key(s->platform_counts, &p);
    return p;
}
const char *dut_bam_stats_infer_platform(const dut_bam_stats *s)
{
    if (!s) return "Unknown";
    const bool any = top_key(s->instruments, &s->top_instrument);
    return specific_platform(dut_bam_stats_primary_platform(s), any ? s->top_instrument.c_str() : nullptr);
}
int dut_bam_stats_get(const dut_bam_stats *s, const char *key, double *out)
{
    if (!s || !key || !out) return 0;
    const sv k(key);
    if (s->read_count > 0) {
        if (k == "average_read_length") { *out = (double)dut_bam_stats_modal_read_length(s); return 1; }
        if (k == "paired_percentage") { *out = ((double)s->paired_reads / (double)s->read_count) * 100.0; return 1; }
    }
    if (s->paired_count > 0) {
        if (k == "average_insert_size") { int64_t m = 0; top_key(s->insert_size_distribution, &m); *out = (double)m; return 1; }
        if (k == "proper_pair_percentage") { *out = ((double)s->paired_count * 2.0 / (double)s->paired_reads) * 100.0; return 1; }
    }
    return 0;
}

size_t dut_format_f64(double v, char *buf)
{
    if (!std::isfinite(v)) { memcpy(buf, "null", 5); return 4; }
    char *o = buf;
    if (std::signbit(v)) { *o++ = '-'; v = -v; }
    if (v == 0.0) { memcpy(o, "0.0", 4); return (size_t)(o - buf) + 3; }
    // shortest round-trip digits, as d[.ddd]e[+-]xx
    char t[40];
    const auto r = std::to_chars(t, t + sizeof(t) - 1, v, std::chars_format::scientific);
    *r.ptr = 0;
    char digits[24]; int n = 0;
    const char *p = t;
    for (; p < r.ptr && *p != 'e'; ++p) if (*p != '.') digits[n++] = *p;
    const int e10 = atoi(p + 1);
    const int kk = e10 + 1;                 // position of the decimal point relative to the first digit
    const int k = kk - n;                   // value = digits * 10^k
    if (k >= 0 && kk <= 16) {
        memcpy(o, digits, n); o += n;
        memset(o, '0', k); o += k;
        *o++ = '.'; *o++ = '0';
    } else if (kk > 0 && kk <= 16) {
        memcpy(o, digits, kk); o += kk;
        *o++ = '.';
        memcpy(o, digits + kk, n - kk); o += n - kk;
    } else if (kk > -5 && kk <= 0) {
        *o++ = '0'; *o++ = '.';
        memset(o, '0', -kk); o += -kk;
        memcpy(o, digits, n); o += n;
    } else {
        *o++ = digits[0];
        if (n > 1) { *o++ = '.'; memcpy(o, digits + 1, n - 1); o += n - 1; }
        *o++ = 'e';
        o += snprintf(o, 8, "%d", kk - 1);
    }
    *o = 0;
    return (size_t)(o - buf);
}

static int dut_coverage_output_json_impl(const dut_contig_stats *stats, const char *const *names,
                             const uint64_t *state_counts, size_t n, const dut_export_meta *meta,
                             char **json, size_t *json_len)
{
    if ((n && (!stats || !names || !state_counts)) || !meta || !json) return CL_ERR_INVALID;
    std::vector<size_t> order(n);
    for (size_t i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return dut_compare_contig_names(names[a], names[b]) < 0; });
    std::vector<dut_contig_stats> so; std::vector<uint64_t> call;
    for (size_t i : order) { so.push_back(stats[i]); call.push_back(state_counts[6 * i + 1]); }
    dut_genome_summary g;
    dut_genome_summary_build(so.data(), call.data(), n, &g);

    auto S = [](const char *p) { return sv(p ? p : ""); };
    Json j;
    j.open('{');
    j.key("export"); j.open('{');
    j.key("summary"); j.open('{');
    j.key("aligner"); j.str(S(meta->aligner));
    j.key("reference_build"); j.str(S(meta->reference_build));
    j.key("sequencing_platform"); j.str(S(meta->sequencing_platform));
    j.key("read_length"); j.u64(meta->read_length);
    j.key("total_bases"); j.u64(g.total_bases);
    j.key("callable_bases"); j.u64(g.callable_bases);
    j.key("callable_percentage"); j.f64(g.callable_percentage);
    j.key("average_depth"); j.f64(g.average_depth);
    j.key("contigs_analyzed"); j.u64(g.contigs_analyzed);
    j.close('}');
    j.key("contigs"); j.open('[');
    for (size_t i : order) {
        dut_contig_derived d;
        dut_contig_derive(&stats[i], &d);
        const uint64_t *c = state_counts + 6 * i;
        j.sep(); j.open('{');
        j.key("name"); j.str(S(names[i]));
        j.key("length"); j.u64(stats[i].length);
        j.key("unique_reads"); j.u64(stats[i].n_reads);
        j.key("coverage_percent"); j.f64(d.coverage_percent);
        j.key("average_depth"); j.f64(d.average_depth);
        j.key("covered_bases"); j.u64(stats[i].n_covered_bases);
        j.key("total_bases"); j.u64(stats[i].length);
        j.key("quality_stats"); j.open('{');
        j.key("average_mapq"); j.f64(d.average_mapq);
        j.key("average_baseq"); j.f64(d.average_baseq);
        j.key("q30_percentage"); j.f64(d.q30_percentage);
        j.close('}');
        j.key("state_distribution"); j.open('{');
        j.key("ref_n"); j.u64(c[0]);
        j.key("callable"); j.u64(c[1]);
        j.key("no_coverage"); j.u64(c[2]);
        j.key("low_coverage"); j.u64(c[3]);
        j.key("excessive_coverage"); j.u64(c[4]);
        j.key("poor_mapping_quality"); j.u64(c[5]);
        j.close('}');
        j.close('}');                     // coverage_histogram is None: skipped (skip_serializing_if)
    }
    j.close(']');
    j.key("quality_metrics"); j.open('{');
    j.key("average_mapq"); j.f64(g.average_mapq);
    j.key("average_baseq"); j.f64(g.average_baseq);
    j.key("q30_percentage"); j.f64(g.q30_percentage);
    j.close('}');
    j.key("total_unique_reads"); j.u64(g.total_unique_reads);
    j.close('}');
    j.key("files"); j.open('{');
    j.key("bed_file"); j.str(S(meta->bed_file));
    j.key("summary_html"); j.str(S(meta->summary_html));
    j.key("coverage_plots"); j.open('[');
    for (size_t i = 0; i < meta->n_coverage_plots; ++i) { j.sep(); j.str(S(meta->coverage_plots[i])); }
    j.close(']');
    j.close('}');
    j.close('}');

    char *out = (char *)malloc(j.s.size() + 1);
    if (!out) return CL_ERR_INVALID;
    memcpy(out, j.s.c_str(), j.s.size() + 1);
    *json = out;
    if (json_len) *json_len = j.s.size();
    return CL_OK;
}

int dut_coverage_output_json(const dut_contig_stats *stats, const char *const *names,
                             const uint64_t *state_counts, size_t n, const dut_export_meta *meta,
                             char **json, size_t *json_len)
{
    // no exception leaves the library through the C ABI
    try { return dut_coverage_output_json_impl(stats, names, state_counts, n, meta, json, json_len); }
    catch (const std::bad_alloc &) { return CL_ERR_NOMEM; }
    catch (...) { return CL_ERR_INVALID; }
}


// summary.html (report.rs:136-340): the sections, rows and number formats of the reference's report from the same
// export -- in this project's own markup and style sheet (the reference's template files are presentation and
// are not reproduced).
static int dut_write_html_report_impl(const dut_contig_stats *stats, const char *const *names, const uint64_t *state_counts,
                          size_t n, const dut_export_meta *meta, uint64_t bam_stats_max_samples, const char *html_path)
{
    if ((n && (!stats || !names || !state_counts)) || !meta || !html_path) return CL_ERR_INVALID;
    std::vector<size_t> order(n);
    for (size_t i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return dut_compare_contig_names(names[a], names[b]) < 0; });
    std::vector<dut_contig_stats> so; std::vector<uint64_t> call;
    for (size_t i : order) { so.push_back(stats[i]); call.push_back(state_counts[6 * i + 1]); }
    dut_genome_summary g;
    dut_genome_summary_build(so.data(), call.data(), n, &g);
    auto esc = [](const char *p) {
        std::string o;
        for (const char *c = p ? p : ""; *c; ++c) {
            switch (*c) { case '&': o += "&amp;"; break; case '<': o += "&lt;"; break; case '>': o += "&gt;"; break;
                          case '"': o += "&quot;"; break; default: o += *c; }
        }
        return o;
    };
    auto fx = [](const char *fmt, double v) { char b[64]; snprintf(b, sizeof(b), fmt, v); return std::string(b); };   // {:.N}
    auto u = [](uint64_t v) { return std::to_string(v); };
    std::string h;
    h += "<!DOCTYPE html>\n<html lang=\"en\"><head><meta charset=\"utf-8\"><title>Coverage summary</title>\n<style>\n"
         "body{font-family:sans-serif;margin:1.5em;color:#222}h1{font-size:1.4em}h2{font-size:1.1em}\n"
         ".stats-box{border:1px solid #ccc;padding:.6em 1em;margin-bottom:1em}.stats-columns{display:flex;gap:3em}\n"
         "dl{display:grid;grid-template-columns:auto auto;gap:.2em 1em;margin:0}dt{font-weight:bold}dd{margin:0}\n"
         "table{border-collapse:collapse;margin:.6em 0}td,th{border:1px solid #ccc;padding:.25em .7em;text-align:left}\n"
         ".group td{font-weight:bold;background:#f5f5f5}.tab-panel{display:none}.tab-panel.active{display:block}\n"
         ".sample-note{font-weight:normal;color:#666;font-size:.85em}figure{margin:.5em 0;overflow-x:auto}\n"
         "</style></head><body>\n<h1>Callable loci coverage summary</h1>\n";
    // ---- BAM statistics (write_bam_stats_section, :162-212) ----
    h += "<section class=\"stats-box\"><h2>BAM Statistics <span class=\"sample-note\">(based on first " + u(bam_stats_max_samples) + " reads)</span></h2>\n";
    h += "<div class=\"stats-columns\"><dl>";
    h += "<dt>Reference Build</dt><dd>" + esc(meta->reference_build) + "</dd>";
    h += "<dt>Aligner</dt><dd>" + esc(meta->aligner) + "</dd>";
    h += "<dt>Sequencing Platform</dt><dd>" + esc(meta->sequencing_platform) + "</dd>";
    h += "<dt>Average read length</dt><dd>" + u(meta->read_length) + " bp</dd>";
    h += "<dt>Total Unique Reads</dt><dd>" + u(g.total_unique_reads) + "</dd>";
    h += "<dt>Total Bases</dt><dd>" + u(g.total_bases) + "</dd></dl>\n<dl>";
    h += "<dt>Callable Bases</dt><dd>" + u(g.callable_bases) + "</dd>";
    h += "<dt>Callable Percentage</dt><dd>" + fx("%.2f", g.callable_percentage) + "%</dd>";
    h += "<dt>Average Depth</dt><dd>" + fx("%.2f", g.average_depth) + "\xC3\x97</dd>";
    h += "<dt>Contigs Analyzed</dt><dd>" + u(g.contigs_analyzed) + "</dd>";
    h += "<dt>Average MapQ</dt><dd>" + fx("%.1f", g.average_mapq) + "</dd>";
    h += "<dt>Average BaseQ</dt><dd>" + fx("%.1f", g.average_baseq) + "</dd></dl></div></section>\n";
    // ---- one panel per contig (write_contig_analysis_section / write_contig_panel, :214-330) ----
    h += "<div class=\"contig-analysis\"><div class=\"contig-selector\"><select id=\"contig-select\" aria-label=\"Select contig\">";
    for (size_t k = 0; k < order.size(); ++k)
        h += "<option value=\"panel-" + u(k) + "\"" + (k == 0 ? " selected" : "") + ">" + esc(names[order[k]]) + "</option>";
    h += "</select></div>\n<div class=\"contig-panels\">\n";
    auto row = [&](const char *label, const std::string &v) { h += std::string("<tr><td>") + label + "</td><td>" + v + "</td></tr>"; };
    for (size_t k = 0; k < order.size(); ++k) {
        const size_t i = order[k];
        dut_contig_derived d;
        dut_contig_derive(&stats[i], &d);
        const uint64_t *c = state_counts + 6 * i;
        h += "<div class=\"tab-panel" + std::string(k == 0 ? " active" : "") + "\" id=\"panel-" + u(k) + "\"><table><thead><tr><th>Metric</th><th>Value</th></tr></thead><tbody>";
        row("Length", u(stats[i].length) + " bp");
        row("Unique Reads", u(stats[i].n_reads));
        row("Covered Bases", u(stats[i].n_covered_bases));
        row("Coverage Percent", fx("%.2f", d.coverage_percent) + "%");
        row("Average Depth", fx("%.2f", d.average_depth) + "\xC3\x97");
        h += "<tr class=\"group\"><td colspan=\"2\">Quality Metrics</td></tr>";
        row("Average MapQ", fx("%.1f", d.average_mapq));
        row("Average BaseQ", fx("%.1f", d.average_baseq));
        row("Q30 Percentage", fx("%.2f", d.q30_percentage) + "%");
        h += "<tr class=\"group\"><td colspan=\"2\">State Distribution</td></tr>";
        row("Reference N", u(c[0])); row("Callable", u(c[1])); row("No Coverage", u(c[2]));
        row("Low Coverage", u(c[3])); row("Excessive Coverage", u(c[4])); row("Poor Mapping Quality", u(c[5]));
        h += "</tbody></table>";
        const std::string plot = std::string(names[i]) + "_coverage.svg";          // looked up relative to the working directory, :316-317
        if (FILE *pf = fopen(plot.c_str(), "rb")) {
            fclose(pf);
            h += "<figure class=\"coverage-plot\"><img src=\"" + esc(plot.c_str()) + "\" alt=\"Coverage distribution for " + esc(names[i]) +
                 "\" loading=\"lazy\"><figcaption>Coverage distribution for " + esc(names[i]) + "</figcaption></figure>";
        }
        h += "</div>\n";
    }
    h += "</div></div>\n<script>\n"
         "document.getElementById('contig-select').addEventListener('change', function () {\n"
         "  var panels = document.querySelectorAll('.tab-panel');\n"
         "  for (var i = 0; i < panels.length; i++) panels[i].classList.toggle('active', panels[i].id === this.value);\n"
         "});\n</script>\n</body></html>\n";
    FILE *f = fopen(html_path, "wb");
    if (!f) return CL_ERR_INVALID;
    fwrite(h.data(), 1, h.size(), f);
    fclose(f);
    return CL_OK;
}

int dut_write_html_report(const dut_contig_stats *stats, const char *const *names, const uint64_t *state_counts,
                          size_t n, const dut_export_meta *meta, uint64_t bam_stats_max_samples, const char *html_path)
{
    // no exception leaves the library through the C ABI
    try { return dut_write_html_report_impl(stats, names, state_counts, n, meta, bam_stats_max_samples, html_path); }
    catch (const std::bad_alloc &) { return CL_ERR_NOMEM; }
    catch (...) { return CL_ERR_INVALID; }
}


void dut_free(void *p) { free(p); }

} // extern "C"
