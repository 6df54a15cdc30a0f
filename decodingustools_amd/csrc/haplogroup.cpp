// haplogroup.cpp -- implementation of include/dut_haplogroup.h: tree JSON -> tree, site list, per-site
// calls, branch scoring and the TSV of `find-y-branch` / `find-mt-branch`.  Host-only except
// dut_find_branch_files, which runs the device engine's cl_site_pileup.
#include "../../include/dut_haplogroup.h"
#include "../../include/dut_report.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <utility>
#include "host_parallel.h"
#include <chrono>
#include <thread>
#include <vector>

namespace {

void set_err(char *err, size_t n, const std::string &m)
{
    if (err && n) snprintf(err, n, "%s", m.c_str());
}

// ---------------------------------------------------------------------------------------------
// a small JSON reader (objects keep document order; a repeated key: the last one wins)
// ---------------------------------------------------------------------------------------------
struct JVal {
    enum Type { Null, Bool, Num, Str, Arr, Obj } t = Null;
    bool b = false;
    bool is_int = false, neg = false;
    uint64_t mag = 0;                 // |value| when is_int
    double num = 0.0;
    std::string s;
    std::vector<JVal> a;
    std::vector<std::pair<std::string, JVal>> o;
    const JVal *get(const char *k) const
    {
        const JVal *r = nullptr;
        for (const auto &kv : o) if (kv.first == k) r = &kv.second;
        return r;
    }
};

struct JParser {
    const char *p, *e;
    std::string err;
    int depth = 0;
    bool fail(const std::string &m) { if (err.empty()) err = m + " at byte " + std::to_string((size_t)(p - start)); return false; }
    const char *start;
    JParser(const char *d, size_t n) : p(d), e(d + n), start(d) {}
    void ws() { while (p < e && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; }
    static void utf8(std::string &o, uint32_t c)
    {
        if (c < 0x80) o += (char)c;
        else if (c < 0x800) { o += (char)(0xC0 | (c >> 6)); o += (char)(0x80 | (c & 0x3F)); }
        else if (c < 0x10000) { o += (char)(0xE0 | (c >> 12)); o += (char)(0x80 | ((c >> 6) & 0x3F)); o += (char)(0x80 | (c & 0x3F)); }
        else { o += (char)(0xF0 | (c >> 18)); o += (char)(0x80 | ((c >> 12) & 0x3F)); o += (char)(0x80 | ((c >> 6) & 0x3F)); o += (char)(0x80 | (c & 0x3F)); }
    }
    bool hex4(uint32_t &v)
    {
        if (e - p < 4) return fail("bad \\u escape");
        v = 0;
        for (int i = 0; i < 4; ++i) {
            const char c = *p++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (uint32_t)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (uint32_t)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (uint32_t)(c - 'A' + 10);
            else return fail("bad \\u escape");
        }
        return true;
    }
    bool str(std::string &o)
    {
        if (p >= e || *p != '"') return fail("expected a string");
        ++p;
        for (;;) {
            if (p >= e) return fail("unterminated string");
            const unsigned char c = (unsigned char)*p++;
            if (c == '"') return true;
            if (c < 0x20) return fail("control character in string");
            if (c != '\\') { o += (char)c; continue; }
            if (p >= e) return fail("unterminated string");
            const char x = *p++;
            switch (x) {
            case '"': o += '"'; break; case '\\': o += '\\'; break; case '/': o += '/'; break;
            case 'b': o += '\b'; break; case 'f': o += '\f'; break; case 'n': o += '\n'; break;
            case 'r': o += '\r'; break; case 't': o += '\t'; break;
            case 'u': {
                uint32_t v = 0;
                if (!hex4(v)) return false;
                if (v >= 0xD800 && v <= 0xDBFF) {
                    uint32_t lo = 0;
                    if (e - p < 2 || p[0] != '\\' || p[1] != 'u') return fail("lone surrogate");
                    p += 2;
                    if (!hex4(lo) || lo < 0xDC00 || lo > 0xDFFF) return fail("lone surrogate");
                    v = 0x10000 + ((v - 0xD800) << 10) + (lo - 0xDC00);
                } else if (v >= 0xDC00 && v <= 0xDFFF) return fail("lone surrogate");
                utf8(o, v);
                break;
            }
            default: return fail("bad escape");
            }
        }
    }
    bool value(JVal &v)
    {
        ws();
        if (p >= e) return fail("unexpected end of input");
        if (++depth > 256) return fail("nesting too deep");
        bool ok = true;
        const char c = *p;
        if (c == '{') {
            v.t = JVal::Obj; ++p; ws();
            if (p < e && *p == '}') ++p;
            else for (v.o.reserve(12);;) {             // the trees' objects have 7-12 members: no regrowth of heavy elements
                ws();
                std::string k;
                if (!str(k)) { ok = false; break; }
                ws();
                if (p >= e || *p != ':') { ok = fail("expected ':'"); break; }
                ++p;
                v.o.emplace_back(std::move(k), JVal());
                if (!value(v.o.back().second)) { ok = false; break; }
                ws();
                if (p < e && *p == ',') { ++p; continue; }
                if (p < e && *p == '}') { ++p; break; }
                ok = fail("expected ',' or '}'"); break;
            }
        } else if (c == '[') {
            v.t = JVal::Arr; ++p; ws();
            if (p < e && *p == ']') ++p;
            else for (v.a.reserve(4);;) {
                v.a.emplace_back();
                if (!value(v.a.back())) { ok = false; break; }
                ws();
                if (p < e && *p == ',') { ++p; continue; }
                if (p < e && *p == ']') { ++p; break; }
                ok = fail("expected ',' or ']'"); break;
            }
        } else if (c == '"') { v.t = JVal::Str; ok = str(v.s); }
        else if (e - p >= 4 && !memcmp(p, "true", 4)) { v.t = JVal::Bool; v.b = true; p += 4; }
        else if (e - p >= 5 && !memcmp(p, "false", 5)) { v.t = JVal::Bool; v.b = false; p += 5; }
        else if (e - p >= 4 && !memcmp(p, "null", 4)) { v.t = JVal::Null; p += 4; }
        else if (c == '-' || (c >= '0' && c <= '9')) {
            const char *q = p;
            if (*q == '-') ++q;
            const char *d0 = q;
            while (q < e && *q >= '0' && *q <= '9') ++q;
            if (q == d0 || (q - d0 > 1 && *d0 == '0')) ok = fail("bad number");
            bool integral = true;
            if (ok && q < e && *q == '.') { integral = false; ++q; const char *f0 = q; while (q < e && *q >= '0' && *q <= '9') ++q; if (q == f0) ok = fail("bad number"); }
            if (ok && q < e && (*q == 'e' || *q == 'E')) { integral = false; ++q; if (q < e && (*q == '+' || *q == '-')) ++q; const char *x0 = q; while (q < e && *q >= '0' && *q <= '9') ++q; if (q == x0) ok = fail("bad number"); }
            if (ok) {
                v.t = JVal::Num;
                v.neg = *p == '-';
                if (integral && (q - d0) <= 19) {
                    // an integer of at most 19 digits fits 64 bits; its nearest double is what strtod would return
                    uint64_t m = 0;
                    for (const char *d = d0; d < q; ++d) m = m * 10u + (uint64_t)(*d - '0');
                    v.is_int = true; v.mag = m;
                    v.num = v.neg ? -(double)m : (double)m;
                } else {
                    const std::string txt(p, q);
                    v.num = strtod(txt.c_str(), nullptr);
                }
                p = q;
            }
        } else ok = fail("unexpected character");
        --depth;
        return ok;
    }
    bool document(JVal &v)
    {
        if (!value(v)) return false;
        ws();
        if (p != e) return fail("trailing characters");
        return true;
    }
};

// serde-style typed access ------------------------------------------------------------------------
bool as_u32(const JVal *v, uint32_t &out) { if (!v || v->t != JVal::Num || !v->is_int || v->neg || v->mag > 0xFFFFFFFFull) return false; out = (uint32_t)v->mag; return true; }
bool as_i32(const JVal *v, int32_t &out)
{
    if (!v || v->t != JVal::Num || !v->is_int) return false;
    if (v->neg ? v->mag > 0x80000000ull : v->mag > 0x7FFFFFFFull) return false;
    out = v->neg ? (int32_t)(-(int64_t)v->mag) : (int32_t)v->mag;
    return true;
}
bool as_str(const JVal *v, std::string &out) { if (!v || v->t != JVal::Str) return false; out = v->s; return true; }
bool as_bool(const JVal *v, bool &out) { if (!v || v->t != JVal::Bool) return false; out = v->b; return true; }

// ---------------------------------------------------------------------------------------------
// tree model (haplogroup/types.rs)
// ---------------------------------------------------------------------------------------------
struct Coord { uint32_t position; std::string chromosome, ancestral, derived; };
struct Locus {
    std::string name;
    bool is_snp = true;                                   // LociType
    std::vector<std::pair<std::string, Coord>> coords;    // build id -> coordinate (unique keys)
    const Coord *get(const std::string &build) const
    {
        for (const auto &c : coords) if (c.first == build) return &c.second;
        return nullptr;
    }
    void put(const std::string &build, Coord c)
    {
        for (auto &e : coords) if (e.first == build) { e.second = std::move(c); return; }
        coords.emplace_back(build, std::move(c));
    }
};
struct Node {                                             // HaplogroupNode
    uint32_t haplogroup_id = 0, parent_id = 0;
    std::string name;
    bool is_root = false;
    std::vector<Locus> loci;
    std::vector<uint32_t> children;
};
struct Haplogroup {
    std::string name;
    bool has_parent = false;
    std::string parent;
    const std::vector<Locus> *loci = nullptr;
    std::vector<Haplogroup> children;
};

using NodeMap = std::map<std::string, Node>;              // all_nodes, keyed by the JSON key / index string

bool build_tree(const NodeMap &all, uint32_t node_id, Haplogroup &out, int depth, bool &too_deep)
{
    if (depth > 100000) { too_deep = true; return false; }
    const auto it = all.find(std::to_string(node_id));
    if (it == all.end()) return false;
    const Node &n = it->second;
    out.name = n.name;
    out.loci = &n.loci;
    for (uint32_t c : n.children) {
        Haplogroup ch;
        if (build_tree(all, c, ch, depth + 1, too_deep)) out.children.push_back(std::move(ch));
        if (too_deep) return false;
    }
    if (n.parent_id != 0) {
        const auto pit = all.find(std::to_string(n.parent_id));
        if (pit == all.end()) return false;               // `?` on the parent lookup: the node is dropped
        out.has_parent = true; out.parent = pit->second.name;
    }
    return true;
}

size_t count_nodes(const Haplogroup &h) { size_t n = 1; for (const auto &c : h.children) n += count_nodes(c); return n; }

bool parse_ftdna(const JVal &doc, NodeMap &all, std::string &err)
{
    const JVal *nodes = doc.t == JVal::Obj ? doc.get("allNodes") : nullptr;
    if (!nodes || nodes->t != JVal::Obj) { err = "missing field `allNodes`"; return false; }
    for (const auto &kv : nodes->o) {
        const JVal &j = kv.second;
        if (j.t != JVal::Obj) { err = "invalid type: expected struct FtdnaNode"; return false; }
        Node n;
        std::string root_s; bool b; uint32_t u;
        if (!as_u32(j.get("haplogroupId"), n.haplogroup_id)) { err = "missing or invalid field `haplogroupId`"; return false; }
        if (j.get("parentId") && !as_u32(j.get("parentId"), n.parent_id)) { err = "invalid field `parentId`"; return false; }
        if (!as_str(j.get("name"), n.name)) { err = "missing or invalid field `name`"; return false; }
        if (!as_bool(j.get("isRoot"), b)) { err = "missing or invalid field `isRoot`"; return false; }
        n.is_root = b;
        if (!as_str(j.get("root"), root_s)) { err = "missing or invalid field `root`"; return false; }
        if (!as_u32(j.get("kitsCount"), u)) { err = "missing or invalid field `kitsCount`"; return false; }
        if (!as_u32(j.get("subBranches"), u)) { err = "missing or invalid field `subBranches`"; return false; }
        if (!as_u32(j.get("bigYCount"), u)) { err = "missing or invalid field `bigYCount`"; return false; }
        if (const JVal *vs = j.get("variants")) {
            if (vs->t != JVal::Arr) { err = "invalid field `variants`"; return false; }
            for (const JVal &v : vs->a) {
                if (v.t != JVal::Obj) { err = "invalid type: expected struct FtdnaVariant"; return false; }
                Locus l;                                   // From<FtdnaVariant> for Locus, ftdna.rs:26-49
                std::string anc, der, tmp;
                if (v.get("variant") && !as_str(v.get("variant"), l.name)) { err = "invalid field `variant`"; return false; }
                if (v.get("ancestral") && !as_str(v.get("ancestral"), anc)) { err = "invalid field `ancestral`"; return false; }
                if (v.get("derived") && !as_str(v.get("derived"), der)) { err = "invalid field `derived`"; return false; }
                if (v.get("region") && !as_str(v.get("region"), tmp)) { err = "invalid field `region`"; return false; }
                if (const JVal *id = v.get("id")) { uint32_t x; if (id->t != JVal::Null && !as_u32(id, x)) { err = "invalid field `id`"; return false; } }
                if (const JVal *pp = v.get("position")) {
                    if (pp->t != JVal::Null) {
                        int32_t pos;
                        if (!as_i32(pp, pos)) { err = "invalid field `position`"; return false; }
                        const uint32_t ap = pos < 0 ? (uint32_t)(-(int64_t)pos) : (uint32_t)pos;     // unsigned_abs
                        l.put("GRCh38", Coord{ap, "chrY", anc, der});
                    }
                }
                n.loci.push_back(std::move(l));
            }
        }
        if (const JVal *cs = j.get("children")) {
            if (cs->t != JVal::Arr) { err = "invalid field `children`"; return false; }
            for (const JVal &c : cs->a) { uint32_t id; if (!as_u32(&c, id)) { err = "invalid field `children`"; return false; } n.children.push_back(id); }
        }
        all[kv.first] = std::move(n);
    }
    return true;
}

bool parse_decodingus(const JVal &doc, NodeMap &all, std::string &err)
{
    if (doc.t != JVal::Arr) { err = "invalid type: expected a sequence"; return false; }
    struct Api { std::string name; bool has_parent = false; std::string parent; const JVal *variants = nullptr; };
    std::vector<Api> api;
    for (const JVal &j : doc.a) {
        if (j.t != JVal::Obj) { err = "invalid type: expected struct ApiNode"; return false; }
        Api a; std::string tmp; bool b;
        if (!as_str(j.get("name"), a.name)) { err = "missing or invalid field `name`"; return false; }
        if (const JVal *pn = j.get("parentName")) {
            if (pn->t == JVal::Str) { a.has_parent = true; a.parent = pn->s; }
            else if (pn->t != JVal::Null) { err = "invalid field `parentName`"; return false; }
        }
        a.variants = j.get("variants");
        if (!a.variants || a.variants->t != JVal::Arr) { err = "missing or invalid field `variants`"; return false; }
        if (!as_str(j.get("lastUpdated"), tmp)) { err = "missing or invalid field `lastUpdated`"; return false; }
        if (!as_bool(j.get("isBackbone"), b)) { err = "missing or invalid field `isBackbone`"; return false; }
        api.push_back(std::move(a));
    }
    std::unordered_map<std::string, uint32_t> name_to_id;
    bool have_root = false; uint32_t root_id = 0;
    for (size_t i = 0; i < api.size(); ++i) {
        name_to_id[api[i].name] = (uint32_t)i;
        if (!api[i].has_parent || api[i].parent.empty()) {
            if (have_root) { err = "Multiple root nodes found in tree"; return false; }
            have_root = true; root_id = (uint32_t)i;
        }
    }
    if (!have_root) { err = "No root node found"; return false; }
    std::vector<Node> nodes(api.size());
    for (size_t i = 0; i < api.size(); ++i) {
        Node &n = nodes[i];
        n.haplogroup_id = (uint32_t)i;
        n.is_root = n.haplogroup_id == root_id;
        n.name = api[i].name;
        if (n.is_root) n.parent_id = 0;
        else if (api[i].has_parent && !api[i].parent.empty()) { const auto it = name_to_id.find(api[i].parent); n.parent_id = it == name_to_id.end() ? root_id : it->second; }
        else n.parent_id = root_id;
        for (const JVal &v : api[i].variants->a) {
            if (v.t != JVal::Obj) { err = "invalid type: expected struct ApiVariant"; return false; }
            Locus l; std::string vt;
            if (!as_str(v.get("name"), l.name)) { err = "missing or invalid field `name`"; return false; }
            if (!as_str(v.get("variantType"), vt)) { err = "missing or invalid field `variantType`"; return false; }
            l.is_snp = vt == "SNP";
            const JVal *co = v.get("coordinates");
            if (!co || co->t != JVal::Obj) { err = "missing or invalid field `coordinates`"; return false; }
            for (const auto &kv : co->o) {
                const JVal &c = kv.second;
                uint32_t start, stop; std::string anc, der;
                if (c.t != JVal::Obj || !as_u32(c.get("start"), start) || !as_u32(c.get("stop"), stop) || !as_str(c.get("anc"), anc) || !as_str(c.get("der"), der)) { err = "invalid ApiCoordinate"; return false; }
                std::string build = kv.first;                                 // accession -> build id, decoding_us.rs:128-135
                if (build == "CM000686.2" || build == "NC_000024.10") build = "GRCh38";
                else if (build == "NC_060948.1" || build == "CP086569.2") build = "T2T-CHM13v2.0";
                else if (build == "CM000686.1") build = "GRCh37";
                l.put(build, Coord{start, build == "GRCh37" ? "Y" : "chrY", anc, der});
            }
            n.loci.push_back(std::move(l));
        }
    }
    for (size_t i = 0; i < nodes.size(); ++i)                                  // children, ascending child index
        if (!nodes[i].is_root && nodes[i].parent_id < nodes.size()) nodes[nodes[i].parent_id].children.push_back((uint32_t)i);
    for (size_t i = 0; i < nodes.size(); ++i) all[std::to_string(i)] = std::move(nodes[i]);
    return true;
}

const char CODE[] = "=ACMGRSVTWYHKDBN";                    // rust-htslib seq().as_bytes()

struct Result {                                             // HaplogroupResult with an owned name pointer into the tree
    const std::string *name;
    double score;
    uint32_t matching, mismatching, ancestral, no_calls, total_snps, cumulative, depth;
};
struct Score { uint32_t matches = 0, ancestral = 0, no_calls = 0, total_snps = 0; double score = 0.0; };

struct Scorer {
    const std::unordered_map<uint32_t, const dut_snp_call *> &calls;
    const std::string &build;
    std::vector<Result> scores;
    std::string err;
    // cumulative_snps (a HashSet cloned per node in the reference, scoring.rs:17-19,124) kept as
    // reference counts along the current root path: its size is `distinct`
    std::unordered_map<uint32_t, uint32_t> path_cnt;
    uint32_t distinct = 0;

    // scoring.rs:8-148.  cumulative = |positions from the root down to and including h|.  false on error.
    bool run(const Haplogroup &h, uint32_t depth, Score &out, uint32_t &cumulative)
    {
        Score cur;
        std::vector<const Coord *> defining;
        for (const Locus &l : *h.loci) if (l.is_snp) if (const Coord *c = l.get(build)) defining.push_back(c);
        for (const Coord *c : defining) if (path_cnt[c->position]++ == 0) ++distinct;
        cumulative = distinct;
        const bool ok = body(h, depth, defining, cur, cumulative);
        for (const Coord *c : defining) if (--path_cnt[c->position] == 0) --distinct;
        out = cur;
        return ok;
    }

    bool body(const Haplogroup &h, uint32_t depth, const std::vector<const Coord *> &defining, Score &cur, uint32_t cumulative)
    {
        int derived = 0, ancestral = 0, no_calls = 0, low_q = 0;
        for (const Coord *c : defining) {
            const auto it = calls.find(c->position);
            if (it == calls.end()) { no_calls += 1; continue; }
            const dut_snp_call &k = *it->second;
            if (k.depth >= 4) {                                                // MIN_DEPTH
                if (c->derived.empty() || c->ancestral.empty()) { err = "locus at position " + std::to_string(c->position) + " has an empty allele"; return false; }
                const char d = c->derived[0], a = c->ancestral[0];
                if (k.base == d) { if (k.freq >= 0.7) derived += 1; else if (k.freq >= 0.5) derived += 1; else low_q += 1; }
                else if (k.base == a) { if (k.freq >= 0.7) ancestral += 1; else low_q += 1; }
                else if (k.freq >= 0.7) derived += 1;
                else low_q += 1;
            } else no_calls += 1;
        }
        const int total_calls = derived + ancestral + low_q;
        if (total_calls > 0) {
            double branch;
            if (ancestral == 0) branch = derived >= 1 ? 3.08 : 1.0;
            else {
                const int d = derived, a = ancestral;
                if (d >= 3 && a <= d / 2) branch = 2.8;
                else if (d >= 2 && a <= d) branch = 2.5;
                else if (d >= 2) branch = 2.0;
                else if (d == 1 && a <= 2) branch = 1.5;
                else if (a > d * 3) branch = 0.0;
                else branch = 1.0;
            }
            const double quality = low_q == 0 ? 1.1 : 0.9;
            cur.score = branch * quality;
        }
        cur.matches += (uint32_t)derived; cur.ancestral += (uint32_t)ancestral; cur.no_calls += (uint32_t)no_calls;
        cur.total_snps += (uint32_t)defining.size();
        if (ancestral > derived * 10) {
            scores.push_back(Result{&h.name, 0.0, (uint32_t)derived, (uint32_t)low_q, (uint32_t)ancestral, (uint32_t)no_calls,
                                    (uint32_t)defining.size(), cumulative, depth});
            return true;
        }
        for (const Haplogroup &ch : h.children) {
            Score cs; uint32_t cc = 0;
            if (!run(ch, depth + 1, cs, cc)) return false;
            scores.push_back(Result{&ch.name, cs.score, cs.matches, (uint32_t)low_q, cs.ancestral, cs.no_calls,
                                    (uint32_t)defining.size(), cc, depth});
        }
        return true;
    }
};

bool find_path(const Haplogroup &h, const std::string &target, std::vector<const std::string *> &path)
{
    if (h.name == target) { path.push_back(&h.name); return true; }
    for (const Haplogroup &c : h.children)
        if (find_path(c, target, path)) { path.push_back(&h.name); return true; }
    return false;
}

void collect_sites(const Haplogroup &h, const std::string &build, const std::string &ref_name, std::map<uint32_t, bool> &out)
{
    for (const Locus &l : *h.loci)
        if (const Coord *c = l.get(build))
            if (l.is_snp) { bool &rel = out[c->position]; rel = rel || c->chromosome == ref_name; }
    for (const Haplogroup &c : h.children) collect_sites(c, build, ref_name, out);
}

} // namespace

struct dut_tree {
    NodeMap all;
    Haplogroup root;
    size_t built = 0;
};

extern "C" {

dut_tree *dut_tree_parse(const char *json, size_t len, int provider, int tree_type, char *err, size_t err_len)
{
    (void)tree_type;
    if (!json) { set_err(err, err_len, "null tree"); return nullptr; }
    JVal doc;
    JParser jp(json, len);
    const auto T0 = std::chrono::steady_clock::now();
    if (!jp.document(doc)) { set_err(err, err_len, "Failed to parse tree: " + jp.err); return nullptr; }
    const auto T1 = std::chrono::steady_clock::now();
    dut_tree *t = new dut_tree();
    std::string e;
    const bool ok = provider == DUT_PROVIDER_DECODINGUS ? parse_decodingus(doc, t->all, e) : parse_ftdna(doc, t->all, e);
    const auto T2 = std::chrono::steady_clock::now();
    if (getenv("DUT_TIMING")) fprintf(stderr, "[dut-timing] tree: json %.0f ms, nodes %.0f ms\n", std::chrono::duration<double>(T1 - T0).count() * 1e3, std::chrono::duration<double>(T2 - T1).count() * 1e3);
    if (!ok) { set_err(err, err_len, "Failed to parse tree: " + e); delete t; return nullptr; }
    // root selection, tree.rs:29-47
    const Node *root = nullptr;
    if (provider == DUT_PROVIDER_DECODINGUS) {
        for (const auto &kv : t->all) if (kv.second.is_root) { root = &kv.second; break; }
        if (!root) { set_err(err, err_len, "No node marked as root found in DecodingUs tree"); delete t; return nullptr; }
    } else {
        size_t n_roots = 0;
        for (const auto &kv : t->all) if (kv.second.parent_id == 0) { if (!root) root = &kv.second; ++n_roots; }
        if (!root) { set_err(err, err_len, "No root node found in FTDNA tree"); delete t; return nullptr; }
        if (n_roots > 1) { set_err(err, err_len, "Multiple root nodes found in FTDNA tree"); delete t; return nullptr; }
    }
    bool too_deep = false;
    if (!build_tree(t->all, root->haplogroup_id, t->root, 0, too_deep)) {
        set_err(err, err_len, too_deep ? "Failed to build tree: the children links form a cycle" : "Failed to build tree");
        delete t; return nullptr;
    }
    t->built = count_nodes(t->root);
    return t;
}

dut_tree *dut_tree_load(const char *json_path, int provider, int tree_type, char *err, size_t err_len)
{
    FILE *f = json_path ? fopen(json_path, "rb") : nullptr;
    if (!f) { set_err(err, err_len, std::string("Failed to get haplogroup tree: cannot open ") + (json_path ? json_path : "(null)")); return nullptr; }
    std::string data;
    char buf[1 << 16];
    size_t g;
    while ((g = fread(buf, 1, sizeof(buf), f)) > 0) data.append(buf, g);
    fclose(f);
    return dut_tree_parse(data.data(), data.size(), provider, tree_type, err, err_len);
}

void dut_tree_free(dut_tree *t) { delete t; }
size_t dut_tree_total_nodes(const dut_tree *t) { return t ? t->all.size() : 0; }
size_t dut_tree_built_nodes(const dut_tree *t) { return t ? t->built : 0; }
const char *dut_tree_root_name(const dut_tree *t) { return t ? t->root.name.c_str() : ""; }

int dut_tree_collect_sites(const dut_tree *t, const char *build_id, const char *ref_name,
                           uint32_t **sites, uint8_t **relevant, size_t *n_sites)
{
    if (!t || !build_id || !ref_name || !sites || !n_sites) return CL_ERR_INVALID;
    std::map<uint32_t, bool> m;
    collect_sites(t->root, build_id, ref_name, m);
    uint32_t *s = (uint32_t *)malloc(std::max<size_t>(m.size(), 1) * sizeof(uint32_t));
    uint8_t *r = (uint8_t *)malloc(std::max<size_t>(m.size(), 1));
    if (!s || !r) { free(s); free(r); return CL_ERR_INVALID; }
    size_t i = 0;
    for (const auto &kv : m) { s[i] = kv.first; r[i] = kv.second ? 1 : 0; ++i; }
    *sites = s; *n_sites = m.size();
    if (relevant) *relevant = r; else free(r);
    return CL_OK;
}

int dut_call_sites(const uint32_t *sites, const uint8_t *relevant, const uint32_t *hist, size_t n_sites,
                   uint32_t min_depth, dut_snp_call **calls, size_t *n_calls)
{
    if ((n_sites && (!sites || !hist)) || !calls || !n_calls) return CL_ERR_INVALID;
    std::vector<dut_snp_call> out;
    for (size_t i = 0; i < n_sites; ++i) {
        if (relevant && !relevant[i]) continue;
        const uint32_t *h = hist + 16 * i;
        uint64_t total = 0; uint32_t best = 0; int bc = 0;
        for (int c = 0; c < 16; ++c) { total += h[c]; if (h[c] > best) { best = h[c]; bc = c; } }
        if (total == 0 || total < min_depth) continue;            // no coverage entry / bases.len() < min_depth
        const double freq = (double)best / (double)(uint32_t)total;
        if (freq >= 0.7) out.push_back(dut_snp_call{sites[i], (uint32_t)total, freq, CODE[bc]});
    }
    std::sort(out.begin(), out.end(), [](const dut_snp_call &a, const dut_snp_call &b) { return a.position < b.position; });
    dut_snp_call *o = (dut_snp_call *)malloc(std::max<size_t>(out.size(), 1) * sizeof(dut_snp_call));
    if (!o) return CL_ERR_INVALID;
    if (!out.empty()) memcpy(o, out.data(), out.size() * sizeof(dut_snp_call));
    *calls = o; *n_calls = out.size();
    return CL_OK;
}

int dut_tree_score(const dut_tree *t, const dut_snp_call *calls, size_t n_calls, const char *build_id,
                   dut_haplogroup_result **results, size_t *n_results, char *err, size_t err_len)
{
    if (!t || (n_calls && !calls) || !build_id || !results || !n_results) return CL_ERR_INVALID;
    std::unordered_map<uint32_t, const dut_snp_call *> cm;
    for (size_t i = 0; i < n_calls; ++i) cm[calls[i].position] = &calls[i];
    const std::string build(build_id);
    Scorer sc{cm, build, {}, {}, {}, 0};
    Score top; uint32_t cum = 0;
    if (!sc.run(t->root, 0, top, cum)) { set_err(err, err_len, sc.err); return CL_ERR_INVALID; }

    // collect_scored_paths, mod.rs:196-258
    std::vector<Result> uniq;
    std::unordered_map<std::string, size_t> at;
    for (const Result &r : sc.scores) {
        const auto it = at.find(*r.name);
        if (it == at.end()) { at[*r.name] = uniq.size(); uniq.push_back(r); }
        else if (r.score > uniq[it->second].score) uniq[it->second] = r;
    }
    std::vector<Result> remaining;
    for (const Result &r : uniq)
        if (r.score > 0.0 && r.ancestral <= r.matching * 3 && r.matching > 0) remaining.push_back(r);
    auto order = [](const Result &a, const Result &b) {
        if (a.cumulative != b.cumulative) return a.cumulative > b.cumulative;
        if (a.score != b.score) return a.score > b.score;
        return *a.name < *b.name;                                   // ties: by name (the reference: hash order)
    };
    std::sort(remaining.begin(), remaining.end(), order);
    std::vector<Result> ordered;
    if (!remaining.empty()) {
        std::vector<const std::string *> path;
        if (find_path(t->root, *remaining.front().name, path)) {
            for (const std::string *nm : path) {
                for (size_t i = 0; i < remaining.size(); ++i)
                    if (*remaining[i].name == *nm) { ordered.push_back(remaining[i]); remaining.erase(remaining.begin() + (long)i); break; }
            }
        }
    }
    std::sort(remaining.begin(), remaining.end(), order);
    ordered.insert(ordered.end(), remaining.begin(), remaining.end());

    dut_haplogroup_result *o = (dut_haplogroup_result *)malloc(std::max<size_t>(ordered.size(), 1) * sizeof(dut_haplogroup_result));
    if (!o) return CL_ERR_INVALID;
    for (size_t i = 0; i < ordered.size(); ++i) {
        const Result &r = ordered[i];
        o[i] = dut_haplogroup_result{r.name->c_str(), r.score, r.matching, r.mismatching, r.ancestral, r.no_calls, r.total_snps, r.cumulative, r.depth};
    }
    *results = o; *n_results = ordered.size();
    return CL_OK;
}

int dut_write_haplogroup_report(const char *path, const dut_tree *t, const dut_haplogroup_result *results,
                                size_t n_results, const dut_snp_call *calls, size_t n_calls,
                                const char *build_id, int show_snps, char *err, size_t err_len)
{
    if (!path || !t || (n_results && !results) || !build_id) return CL_ERR_INVALID;
    std::unordered_map<uint32_t, const dut_snp_call *> cm;
    for (size_t i = 0; i < n_calls; ++i) cm[calls[i].position] = &calls[i];
    const std::string build(build_id);
    // find_haplogroup (mod.rs:183-194) returns the first node of that name in pre-order: one traversal
    // builds the same mapping for all rows
    std::unordered_map<std::string, const Haplogroup *> by_name;
    if (show_snps) {
        std::vector<const Haplogroup *> stack{&t->root};
        while (!stack.empty()) {
            const Haplogroup *h = stack.back(); stack.pop_back();
            by_name.emplace(h->name, h);                      // keeps the first (pre-order) occurrence
            for (size_t i = h->children.size(); i-- > 0;) stack.push_back(&h->children[i]);
        }
    }
    std::string s = "Haplogroup\tScore\tMatching_SNPs\tMismatching_SNPs\tAncestral_Matches\tNo_Calls\tTotal_SNPs\tCumulative_SNPs\tDepth";
    if (show_snps) s += "\tMatching_SNP_Details\tMismatching_SNP_Details\tNo_Call_Details";
    s += "\n";
    for (size_t i = 0; i < n_results; ++i) {
        const dut_haplogroup_result &r = results[i];
        char b[256];
        snprintf(b, sizeof(b), "\t%.4f\t%u\t%u\t%u\t%u\t%u\t%u\t%u", r.score, r.matching_snps, r.mismatching_snps, r.ancestral_matches,
                 r.no_calls, r.total_snps, r.cumulative_snps, r.depth);
        s += r.name; s += b;
        if (show_snps) {                                            // get_snp_details, mod.rs:143-181
            std::string m, mm, nc;
            const auto hit = by_name.find(r.name);
            if (const Haplogroup *h = hit == by_name.end() ? nullptr : hit->second) {
                for (const Locus &l : *h->loci) {
                    const Coord *c = l.get(build);
                    if (!c) continue;
                    const std::string item = l.name + ":" + std::to_string(c->position);
                    const auto it = cm.find(c->position);
                    std::string *dst = &nc;
                    if (it != cm.end()) {
                        if (c->derived.empty()) { set_err(err, err_len, "locus " + l.name + " has an empty derived allele"); return CL_ERR_INVALID; }
                        dst = it->second->base == c->derived[0] ? &m : &mm;
                    }
                    if (!dst->empty()) *dst += ";";
                    *dst += item;
                }
            }
            s += "\t" + m + "\t" + mm + "\t" + nc;
        }
        s += "\n";
    }
    FILE *f = fopen(path, "wb");
    if (!f) { set_err(err, err_len, std::string("cannot create ") + path); return CL_ERR_INVALID; }
    fwrite(s.data(), 1, s.size(), f);
    fclose(f);
    return CL_OK;
}

int dut_validate_reference(const char *header_text, size_t len, const char *const *ref_names, size_t n_refs,
                           int tree_type, char *build_id, size_t build_len, char *chromosome, size_t chrom_len,
                           char *err, size_t err_len)
{
    const std::string genome = dut_reference_build(header_text, len);
    if (genome == "Unknown") { set_err(err, err_len, "Could not determine reference genome from BAM header"); return CL_ERR_INVALID; }
    std::vector<const char *> cand;
    if (tree_type == DUT_TREE_MTDNA) cand = {"chrM", "MT", "M"};
    else if (genome == "GRCh38") cand = {"chrY", "Y", "NC_000024.10", "CM000686.2"};
    else if (genome == "GRCh37") cand = {"Y", "chrY"};
    else if (genome == "T2T-CHM13v2.0") cand = {"Y", "chrY", "CP086569.2", "NC_060948.1"};
    for (const char *c : cand)
        for (size_t i = 0; i < n_refs; ++i)
            if (ref_names[i] && strcmp(ref_names[i], c) == 0) {
                if (build_id && build_len) snprintf(build_id, build_len, "%s", tree_type == DUT_TREE_MTDNA ? "rCRS" : genome.c_str());
                if (chromosome && chrom_len) snprintf(chromosome, chrom_len, "%s", c);
                return CL_OK;
            }
    std::string tried;
    for (size_t i = 0; i < cand.size(); ++i) { if (i) tried += ", "; tried += cand[i]; }
    set_err(err, err_len, "No valid sequence found in BAM. Tried: " + tried);
    return CL_ERR_INVALID;
}

static int dut_find_branch_files_impl(const char *bam_path, const char *fasta_path, const char *tree_json_path,
                          const char *output_path, uint32_t min_depth, uint8_t min_quality, int tree_type,
                          int provider, int show_snps, int device_id, char *err, size_t err_len)
{
    if (!bam_path || !fasta_path || !tree_json_path || !output_path) { set_err(err, err_len, "null argument"); return CL_ERR_INVALID; }
    char e[512] = {0};
    dut_fasta *fa = dut_fasta_open(fasta_path, e, sizeof(e));                 // mod.rs:28
    if (!fa) { set_err(err, err_len, e); return CL_ERR_INVALID; }
    dut_bam *bam = dut_bam_open(bam_path, e, sizeof(e));                      // mod.rs:46 (IndexedReader)
    if (!bam) { dut_fasta_close(fa); set_err(err, err_len, e); return CL_ERR_INVALID; }
    int rc = CL_OK;
    dut_tree *tree = nullptr;
    cl_ctx *ctx = nullptr;
    uint32_t *sites = nullptr; uint8_t *rel = nullptr; size_t n_sites = 0;
    dut_snp_call *calls = nullptr; size_t n_calls = 0;
    dut_haplogroup_result *res = nullptr; size_t n_res = 0;
    char build[64] = {0}, chrom[256] = {0};
    std::vector<const char *> names;
    std::vector<uint32_t> hist;
    int tid = -1;
    if (!dut_bam_has_index(bam)) { set_err(err, err_len, std::string("no .bai or .csi index beside ") + bam_path); rc = CL_ERR_INVALID; goto out; }
    for (int t = 0; t < dut_bam_n_ref(bam); ++t) names.push_back(dut_bam_ref_name(bam, t));
    {
        size_t tl = 0;
        const char *text = dut_bam_header_text(bam, &tl);
        // the reference tests the header text *with* its @SQ lines (HeaderView::as_bytes)
        rc = dut_validate_reference(text, tl, names.data(), names.size(), tree_type, build, sizeof(build), chrom, sizeof(chrom), err, err_len);
        if (rc != CL_OK) goto out;
    }
    for (size_t i = 0; i < names.size(); ++i) if (strcmp(names[i], chrom) == 0) { tid = (int)i; break; }
    {
        // Three independent pieces of work side by side: the tree JSON (one thread: parse + collect the sites),
        // the HIP runtime + engine context (one thread), and the contig's records with their 4-bit sequences
        // (this thread and the decode pool).  Errors are reported in the reference's order: tree, then BAM.
        char terr[512] = {0};
        int trc = CL_OK, crc = CL_OK;
        cl_options opt = {4, 500, 10, 20, 10, 1, 0.1};
        dut::Thread tt = dut::spawn_or_run([&]() {
            tree = dut_tree_load(tree_json_path, provider, tree_type, terr, sizeof(terr));
            if (!tree) { trc = CL_ERR_INVALID; return; }
            trc = dut_tree_collect_sites(tree, build, chrom, &sites, &rel, &n_sites);
            if (trc != CL_OK) snprintf(terr, sizeof(terr), "collect_snps failed");
        });
        dut::Thread ct = dut::spawn_or_run([&]() { crc = cl_create(&opt, device_id, nullptr, &ctx); });
        dut_records rec; const uint64_t *seq_off = nullptr; const uint8_t *seq4 = nullptr;
        const int brc = dut_bam_read_contig(bam, tid, &rec, &seq_off, &seq4);
        const uint8_t *bases = nullptr; uint64_t blen = 0;
        const int frc = dut_fasta_fetch(fa, chrom, &bases, &blen);       // fetch_seq(..)?, caller.rs:110
        if (tt.joinable()) tt.join();
        if (ct.joinable()) ct.join();
        if (frc != CL_OK) { set_err(err, err_len, dut_fasta_error(fa)); rc = frc; goto out; }
        if (trc != CL_OK) { set_err(err, err_len, terr); rc = trc; goto out; }
        if (brc != CL_OK) { set_err(err, err_len, dut_bam_error(bam)); rc = brc; goto out; }
        if (crc != CL_OK) { set_err(err, err_len, "no usable HIP device (the engine has no CPU fallback)"); rc = crc; goto out; }
        cl_site_tile tile;
        tile.n_reads = rec.n; tile.pos = rec.pos; tile.mapq = rec.mapq; tile.cigar_off = rec.cigar_off; tile.cigar = rec.cigar;
        tile.seq_off = seq_off; tile.seq4 = seq4;
        hist.assign(std::max<size_t>(n_sites, 1) * 16, 0u);
        if (n_sites) {
            rc = cl_site_pileup(ctx, min_quality, dut_bam_ref_len(bam, tid), blen, &tile, sites, n_sites, hist.data());
            if (rc != CL_OK) { const char *m = cl_last_error(ctx); set_err(err, err_len, (m && *m) ? m : "site pileup failed"); goto out; }
        }
    }
    rc = dut_call_sites(sites, rel, hist.data(), n_sites, min_depth, &calls, &n_calls);
    if (rc != CL_OK) { set_err(err, err_len, "calling failed"); goto out; }
    rc = dut_tree_score(tree, calls, n_calls, build, &res, &n_res, err, err_len);
    if (rc != CL_OK) goto out;
    rc = dut_write_haplogroup_report(output_path, tree, res, n_res, calls, n_calls, build, show_snps, err, err_len);
out:
    free(res); free(calls); free(sites); free(rel);
    if (ctx) cl_destroy(ctx);
    if (tree) dut_tree_free(tree);
    dut_bam_close(bam);
    dut_fasta_close(fa);
    return rc;
}

int dut_find_branch_files(const char *bam_path, const char *fasta_path, const char *tree_json_path,
                          const char *output_path, uint32_t min_depth, uint8_t min_quality, int tree_type,
                          int provider, int show_snps, int device_id, char *err, size_t err_len)
{
    // no exception leaves the library through the C ABI
    try { return dut_find_branch_files_impl(bam_path, fasta_path, tree_json_path, output_path, min_depth, min_quality, tree_type, provider, show_snps, device_id, err, err_len); }
    catch (const std::bad_alloc &) { set_err(err, err_len, "out of memory or internal error"); return CL_ERR_NOMEM; }
    catch (...) { set_err(err, err_len, "out of memory or internal error"); return CL_ERR_INVALID; }
}


} // extern "C"
