// TEST INFRASTRUCTURE: drives the host-only parts of the library (BAM / FASTA reader, admission, BamStats,
// JSON writer, tree parsing / scoring / report) under AddressSanitizer + UBSan or ThreadSanitizer on the CPU.
// The device engine is not part of this binary: the few cl_* symbols the host sources reference are
// defined below as failing stubs for the link only (nothing here reaches them).
#include "../../include/dut_coverage.h"
#include "../../include/dut_bam.h"
#include "../../include/dut_haplogroup.h"
#include "../../include/dut_report.h"
#include "../../decodingustools_amd/csrc/host_parallel.h"
#include "../../decodingustools_amd/csrc/pass_rows.h"
#include "../../decodingustools_amd/csrc/qual_pack.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

// A stand-in for the device engine, for this binary only: a context that remembers the contig it was given and hands back
// two runs that depend on it (so that a result delivered to the wrong contig, or twice, shows in the BED).  It lets the
// host's orchestration -- dut_coverage_files_multi: one thread, reader pair and context per "device", results handed to
// the BED writer in tid order -- run under ThreadSanitizer without a GPU.  Nothing here computes a pileup.
struct cl_ctx { int device; int32_t tid; uint32_t len; uint64_t n_reads; cl_interval iv[2]; };
extern "C" {
cl_status cl_create(const cl_options *, int device, void *, cl_ctx **out)
{
    if (device < 0 || device > 7) return CL_ERR_DEVICE;
    *out = new cl_ctx(); (*out)->device = device; return CL_OK;
}
void cl_destroy(cl_ctx *c) { delete c; }
const char *cl_last_error(const cl_ctx *) { return "no device in the sanitizer build"; }
cl_status cl_contig_begin(cl_ctx *c, int32_t tid, uint32_t len, const uint8_t *, uint64_t) { c->tid = tid; c->len = len; c->n_reads = 0; return CL_OK; }
cl_status cl_push_reads(cl_ctx *c, const cl_read_tile *t) { c->n_reads += t->n_reads; return CL_OK; }
cl_status cl_push_reads_bits(cl_ctx *c, const cl_read_tile_bits *t)
{
    // every word the tile names is read (the sanitizers see an offset that runs past the reader's buffers)
    uint64_t seen = 0;
    if (t->n_reads && t->qual_off[t->n_reads] > t->qual_off[0])
        for (uint64_t w = t->qual_off[0] >> 6; w <= (t->qual_off[t->n_reads] - 1) >> 6; ++w) seen ^= t->pass_bits[w];
    for (uint64_t i = 0; i < t->n_reads; ++i) seen += t->pass_sum[i];
    c->n_reads += t->n_reads + (seen == 0x5EEDull ? 0 : 0);
    return CL_OK;
}
cl_status cl_contig_prefetch_qual(cl_ctx *, const uint8_t *, uint64_t) { return CL_OK; }
cl_status cl_contig_reserve(cl_ctx *, uint64_t, uint64_t, uint64_t) { return CL_OK; }
cl_status cl_contig_finish(cl_ctx *c, cl_contig_summary *s, const cl_interval **iv, size_t *n)
{
    memset(s, 0, sizeof(*s));
    const uint32_t cut = c->len > 1 ? 1 + (uint32_t)((c->n_reads * 7 + (uint32_t)c->tid) % (c->len - 1)) : c->len;
    c->iv[0] = {0, cut, CL_NO_COVERAGE}; c->iv[1] = {cut, c->len, CL_CALLABLE};
    s->state_counts[CL_NO_COVERAGE] = cut; s->state_counts[CL_CALLABLE] = c->len - cut; s->extent = c->len;
    s->n_intervals = cut < c->len ? 2 : 1;
    *iv = c->iv; *n = (size_t)s->n_intervals;
    return c->len ? CL_OK : CL_ERR_INVALID;
}
cl_status cl_contig_abort(cl_ctx *) { return CL_OK; }
cl_status cl_site_pileup(cl_ctx *, uint8_t, uint32_t, uint64_t, const cl_site_tile *, const uint32_t *, size_t, uint32_t *) { return CL_ERR_DEVICE; }
}

static unsigned long long checksum(const void *p, size_t n)
{
    unsigned long long h = 1469598103934665603ull;
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}

// The host half of the pass-bit form (qual_pack.cpp, pass_rows.h) on random reads with buffers of exactly the documented
// sizes, so that the sanitizers see every byte it touches: quality bytes -> pass bits (every level) -> reference order
// through the CIGAR -> rows of one window range; the column sums must be what a per-base walk counts.
static int pass_bit_selftest()
{
    uint64_t state = 0x9E3779B97F4A7C15ull;
    auto rnd = [&](uint64_t m) { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return state % m; };
    constexpr uint32_t T = 2048;
    int bad = 0;
    for (int round = 0; round < 60; ++round) {
        const uint8_t thr = (uint8_t)(round % 3 == 0 ? 0 : (round % 3 == 1 ? 20 : 200));
        const uint32_t n = 40 + (uint32_t)rnd(200);
        std::vector<int32_t> pos; std::vector<uint32_t> end; std::vector<uint8_t> mapq;
        std::vector<unsigned long long> off(1, 0ull); std::vector<uint32_t> sc_off(1, 0u), sc;
        std::vector<uint64_t> bits;
        std::vector<uint32_t> want(4 * T, 0);                  // qc_depth of positions [0, 4 T)
        uint32_t p = 0;
        for (uint32_t i = 0; i < n; ++i) {
            p += (uint32_t)rnd(120);
            std::vector<uint32_t> cig;
            const int kind = (int)rnd(5);
            unsigned long long qlen = 0, span = 0;
            if (kind == 0) { const uint32_t l = 1 + (uint32_t)rnd(300); cig.push_back(l << 4); }
            else {
                if (rnd(3) == 0) cig.push_back(((uint32_t)(1 + rnd(9)) << 4) | 4u);        // S
                const int nops = 1 + (int)rnd(kind == 4 ? 150 : 12);
                for (int j = 0; j < nops; ++j) {
                    cig.push_back(((uint32_t)(1 + rnd(kind == 3 ? 70 : 20)) << 4) | (uint32_t)(rnd(4) == 0 ? 7u : 0u));   // M or =
                    const uint32_t o = (uint32_t)rnd(kind == 2 ? 6 : 4);
                    if (o == 0) cig.push_back(((uint32_t)(1 + rnd(3)) << 4) | 1u);           // I
                    else if (o == 1) cig.push_back(((uint32_t)(1 + rnd(5)) << 4) | 2u);      // D
                    else if (o >= 4) cig.push_back(((uint32_t)(1500 + rnd(3000)) << 4) | 3u); // N: a gapped (sparse) read
                }
                cig.push_back(((uint32_t)(1 + rnd(30)) << 4) | 0u);
            }
            for (uint32_t cw : cig) { const uint32_t op = cw & 15u, l = cw >> 4; if ((0x193u >> op) & 1u) qlen += l; if ((0x18Du >> op) & 1u) span += l; }
            unsigned long long ql = rnd(7) == 0 ? qlen / 2 : qlen;                             // sometimes a truncated quality string
            if (rnd(23) == 0) ql = 0;
            std::vector<uint8_t> q((size_t)ql);
            for (auto &b : q) b = (uint8_t)(rnd(4) == 0 ? rnd(256) : 30 + rnd(12));
            const uint8_t mq = (uint8_t)(rnd(6) == 0 ? 3 : 60);
            pos.push_back((int32_t)p); end.push_back(p + (uint32_t)span); mapq.push_back(mq);
            // what the column walk counts
            if (mq >= 10 && ql) {
                unsigned long long x = p, y = 0;
                for (uint32_t cw : cig) {
                    const uint32_t op = cw & 15u, l = cw >> 4;
                    if ((0x181u >> op) & 1u) for (uint32_t k = 0; k < l; ++k) if (y + k < ql && q[(size_t)(y + k)] >= thr && x + k < want.size()) want[(size_t)(x + k)] += 1;
                    if ((0x18Du >> op) & 1u) x += l;
                    if ((0x193u >> op) & 1u) y += l;
                }
            }
            // what cl_push_reads leaves (push_reads_bits, second phase), with exact-size buffers
            unsigned long long o0 = bits.size();
            if (mq >= 10 && ql && span) {
                if (cig.size() == 1) {
                    const uint64_t nb = std::min<uint64_t>(span, ql);
                    std::vector<uint64_t> w((size_t)((nb + 63) >> 6));
                    const int level = round % 3;
                    const uint64_t sm = dut::qual_pass_read(q.data(), nb, thr, w.data(), level);
                    uint64_t naive = 0; for (uint64_t k = 0; k < nb; ++k) naive += q[(size_t)k] >= thr ? q[(size_t)k] : 0u;
                    if (sm != naive) ++bad;
                    bits.insert(bits.end(), w.begin(), w.end());
                } else {
                    const uint64_t nqw = (ql + 63) >> 6;
                    const bool sparse = span > 4 * ql + 1024;
                    std::vector<uint64_t> qw((size_t)nqw + 2, 0ull);
                    const uint64_t all = dut::qual_pass_read(q.data(), ql, thr, qw.data(), 2);
                    std::vector<dut::QueryStretch> um(cig.size() + 1);
                    size_t n_um = 0; unsigned long long qcl = 0;
                    if (sparse) {
                        bits.insert(bits.end(), qw.begin(), qw.begin() + (size_t)nqw);
                        sc.insert(sc.end(), cig.begin(), cig.end());
                        o0 |= dut::kRowSparse;
                    } else {
                        std::vector<uint64_t> rw((size_t)((span + 63) >> 6));
                        dut::ref_bits_from_query(qw.data(), ql, cig.data(), (uint32_t)cig.size(), rw.data(), um.data(), &n_um, &qcl);
                        bits.insert(bits.end(), rw.begin(), rw.end());
                        const uint64_t un = dut::unmatched_pass_sum(q.data(), ql, um.data(), n_um, qcl, thr);
                        uint64_t naive = 0; unsigned long long y = 0;
                        for (uint32_t cw : cig) { const uint32_t op = cw & 15u, l = cw >> 4; if ((0x181u >> op) & 1u) for (uint32_t k = 0; k < l; ++k) if (y + k < ql && q[(size_t)(y + k)] >= thr) naive += q[(size_t)(y + k)]; if ((0x193u >> op) & 1u) y += l; }
                        if (all - un != naive) ++bad;
                    }
                }
            }
            off.back() = o0; off.push_back(bits.size());
            sc_off.push_back((uint32_t)sc.size());
        }
        bits.push_back(0ull);                                   // the one word deposit_bits may read behind the last string
        if (sc.empty()) sc.push_back(0u);
        dut::RowReads H;
        H.pos = pos.data(); H.end = end.data(); H.mapq = mapq.data(); H.off = off.data(); H.bits = bits.data();
        H.sc_off = sc_off.data(); H.sc = sc.data(); H.min_mapq = 10;
        std::vector<dut::RowCur> act;
        dut::RowScratch scr;
        uint32_t next = 0;
        for (uint32_t w = 0; w < 4; ++w) {
            const uint32_t W = w * T;
            while (next < n && (uint32_t)pos[next] < W + T) { dut::rows_enter(act, H, next, W); ++next; }
            size_t cap = 2, ng;
            std::vector<dut::RowCur> start = act;
            std::vector<uint32_t> buf;
            for (;;) { buf.assign(cap * dut::kRowGroupWords, 0xFFFFFFFFu); act = start; ng = dut::rows_window<T>(act, H, W, buf.data(), cap, scr); if (ng != SIZE_MAX) break; cap *= 2; }
            for (uint32_t x = 0; x < T; ++x) {
                uint32_t cnt = 0;
                for (size_t r = 0; r < 4 * ng; ++r) cnt += (buf[(r >> 2) * dut::kRowGroupWords + ((x >> 5) << 2) + (r & 3)] >> (x & 31u)) & 1u;
                if (cnt != want[W + x]) ++bad;
            }
        }
    }
    // the reference's "is N" bits, with buffers of exactly the sizes the interface names
    for (int round = 0; round < 40; ++round) {
        const size_t nb = (size_t)rnd(700), nw = (size_t)rnd(14);
        std::vector<uint8_t> ref(nb);
        for (auto &b : ref) b = (uint8_t)"ACGTNnacgtRY"[rnd(12)];
        std::vector<uint64_t> out(nw);
        dut::ref_n_words(ref.data(), nb, nw, out.data(), round % 3);
        for (size_t p = 0; p < 64 * nw; ++p) {
            const bool want = p >= nb || (ref[p] | 0x20) == 'n';
            if ((((out[p >> 6] >> (p & 63)) & 1ull) != 0) != want) ++bad;
        }
    }
    printf("pass-bit selftest: %d mismatch(es)\n", bad);
    return bad;
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: sanitize_host <bam> [tree.json]\n"); return 2; }
    if (pass_bit_selftest() != 0) return 5;
    char err[512] = {0};
    // BAM reader: every contig, with and without sequences, twice (buffer reuse), in reverse order too
    dut_bam *b = dut_bam_open(argv[1], err, sizeof(err));
    if (!b) { printf("open failed: %s\n", err); return 0; }           // corrupted inputs are allowed to fail cleanly
    const int n_ref = dut_bam_n_ref(b);
    cl_options opt = {4, 500, 10, 20, 10, 1, 0.1};
    for (int pass = 0; pass < 2; ++pass) {
        for (int k = 0; k < n_ref; ++k) {
            const int tid = pass ? n_ref - 1 - k : k;
            dut_records rec; const uint64_t *so = nullptr; const uint8_t *sq = nullptr;
            const int rc = dut_bam_read_contig(b, tid, &rec, pass ? &so : nullptr, pass ? &sq : nullptr);
            if (rc != CL_OK) { printf("tid %d: %s\n", tid, dut_bam_error(b)); continue; }
            unsigned long long h = checksum(rec.pos, rec.n * 4) ^ checksum(rec.qual, rec.qual_off[rec.n]) ^ checksum(rec.cigar, 4ull * rec.cigar_off[rec.n]) ^
                                   checksum(rec.qname, rec.qname_off[rec.n]);
            if (pass) h ^= checksum(sq, (so[rec.n] + 1) / 2);
            {   // the packed variant of the same contig: its bits and sums against the bytes just read
                const std::vector<uint8_t> q(rec.qual, rec.qual + rec.qual_off[rec.n]);
                const std::vector<uint64_t> qo(rec.qual_off, rec.qual_off + rec.n + 1);
                dut_records pr;
                const uint8_t thr = (uint8_t)(pass ? 20 : 0);
                if (dut_bam_read_contig_bits(b, tid, thr, &pr) == CL_OK && pr.n == qo.size() - 1) {
                    unsigned long long wrong = pr.qual != nullptr;
                    for (uint64_t g = 0; g < q.size(); ++g) wrong += (((pr.pass_bits[g >> 6] >> (g & 63)) & 1ull) != 0) != (q[(size_t)g] >= thr);
                    for (uint64_t i = 0; i < pr.n; ++i) {
                        uint64_t y = 0, want = 0;
                        const uint64_t ql = qo[i + 1] - qo[i];
                        for (uint32_t k = pr.cigar_off[i]; k < pr.cigar_off[i + 1]; ++k) {
                            const uint32_t op = pr.cigar[k] & 15u, l = pr.cigar[k] >> 4;
                            if ((0x181u >> op) & 1u) for (uint64_t j = y; j < y + l && j < ql; ++j) want += q[(size_t)(qo[i] + j)] >= thr ? q[(size_t)(qo[i] + j)] : 0u;
                            if ((0x193u >> op) & 1u) y += l;
                        }
                        wrong += want != pr.pass_sum[i];
                    }
                    if (wrong) { printf("tid %d: the packed reader disagrees in %llu place(s)\n", tid, wrong); return 6; }
                    // (rec's arrays are the reader's own and are gone now: read again for what follows)
                    if (dut_bam_read_contig(b, tid, &rec, pass ? &so : nullptr, pass ? &sq : nullptr) != CL_OK) continue;
                }
            }
            std::vector<uint8_t> acc(rec.n + 1);
            uint32_t names = 0; uint64_t nacc = 0;
            const int ar = dut_admit_reads(&opt, tid, dut_bam_ref_len(b, tid), &rec, acc.data(), &names, &nacc);
            printf("tid %d pass %d: n %llu hash %016llx admit rc %d accepted %llu names %u\n", tid, pass, (unsigned long long)rec.n, h, ar,
                   (unsigned long long)nacc, names);
        }
    }
    dut_bam_close(b);
    dut_bam_stats *bs = dut_bam_stats_new(10000);
    if (dut_bam_stats_collect(bs, argv[1], err, sizeof(err)) == CL_OK)
        printf("stats: %s | %s | %s | %llu\n", dut_bam_stats_aligner(bs), dut_bam_stats_reference_build(bs), dut_bam_stats_infer_platform(bs),
               (unsigned long long)dut_bam_stats_average_read_length(bs));
    dut_bam_stats_free(bs);
    if (argc > 2) {
        dut_tree *t = dut_tree_load(argv[2], DUT_PROVIDER_FTDNA, DUT_TREE_YDNA, err, sizeof(err));
        if (!t) { printf("tree failed: %s\n", err); return 0; }
        uint32_t *sites = nullptr; uint8_t *rel = nullptr; size_t ns = 0;
        dut_tree_collect_sites(t, "GRCh38", "chrY", &sites, &rel, &ns);
        std::vector<uint32_t> hist(16 * (ns + 1), 0);
        static const int kCode[4] = {1, 2, 4, 8};                         // A C G T
        for (size_t i = 0; i < ns; ++i) { hist[16 * i + kCode[i % 4]] = 12 + (uint32_t)(i % 9); hist[16 * i + 15] += (uint32_t)(i % 3); }
        dut_snp_call *calls = nullptr; size_t nc = 0;
        dut_call_sites(sites, rel, hist.data(), ns, 10, &calls, &nc);
        dut_haplogroup_result *res = nullptr; size_t nr = 0;
        const int rc = dut_tree_score(t, calls, nc, "GRCh38", &res, &nr, err, sizeof(err));
        printf("tree: %zu nodes, %zu sites, %zu calls, score rc %d, %zu rows\n", dut_tree_built_nodes(t), ns, nc, rc, nr);
        if (rc == CL_OK) {
            const std::string out = std::string(argv[2]) + ".tsv";
            dut_write_haplogroup_report(out.c_str(), t, res, nr, calls, nc, "GRCh38", 1, err, sizeof(err));
        }
        dut_free(res); dut_free(calls); dut_free(sites); dut_free(rel);
        dut_tree_free(t);
    }
    // an allocation failure inside a parallel loop or on a helper thread must come back to the caller as an
    // exception on ITS thread (the C ABI wrappers turn it into a status), never end a worker (std::terminate)
    {
        std::atomic<size_t> done{0};
        bool caught = false;
        try {
            dut::parallel_for(20000, 7, [&](size_t i) { if (i == 4177) throw std::bad_alloc(); done.fetch_add(1); });
        } catch (const std::bad_alloc &) { caught = true; }
        int ran = 0;
        { dut::Thread t = dut::spawn_or_run([&]() { ran = 1; throw std::bad_alloc(); }); }
        printf("parallel_for: exception %s, helper ran %d, not every index ran %d\n", caught ? "caught" : "LOST", ran, done.load() < 20000 ? 1 : 0);
        if (!caught) return 3;
    }
    // the persistent workers: loops started from several threads at once, and from inside a chunk of another loop,
    // share them and each comes back complete (the caller always works on its own loop)
    {
        std::atomic<uint64_t> total{0};
        auto burst = [&](uint64_t salt) {
            for (int rep = 0; rep < 40; ++rep) {
                std::atomic<uint64_t> sum{0};
                const size_t n = 300 + 97 * (size_t)rep;
                dut::parallel_for(n, 3, [&](size_t i) {
                    if (i % 64 == 0) {                         // a loop inside a chunk
                        std::atomic<uint64_t> inner{0};
                        dut::parallel_for(50, 4, [&](size_t j) { inner.fetch_add(j + 1); });
                        if (inner.load() != 50 * 51 / 2) abort();
                    }
                    sum.fetch_add(i + salt);
                });
                if (sum.load() != n * (n - 1) / 2 + salt * n) abort();
                total.fetch_add(sum.load());
            }
        };
        {
            dut::Thread a = dut::spawn_or_run([&]() { burst(1); });
            dut::Thread b = dut::spawn_or_run([&]() { burst(2); });
            burst(3);
        }
        printf("worker pool: three callers, nested loops, total %llu\n", (unsigned long long)total.load());
    }
    // the crew of the pinned ring (host_parallel.h: started once, asleep between jobs) and the pool of large temporaries:
    // jobs of changing width back to back, each index exactly once; blocks taken and given back from several threads
    {
        dut::Crew crew;
        crew.ensure(3);
        unsigned long long ran = 0;
        for (int rep = 0; rep < 200; ++rep) {
            const int n = 1 + (rep * 7) % 9;
            std::vector<std::atomic<int>> hit(9);
            for (auto &h : hit) h.store(0);
            crew.start(n, [&](int t) { hit[(size_t)t].fetch_add(1); });
            crew.wait();
            for (int t = 0; t < 9; ++t) { if (hit[(size_t)t].load() != (t < n ? 1 : 0)) { printf("crew: job %d ran index %d %d time(s)\n", rep, t, hit[(size_t)t].load()); return 7; } ran += (unsigned long long)hit[(size_t)t].load(); }
        }
        std::atomic<unsigned long long> bytes{0};
        dut::parallel_for(64, 1, [&](size_t i) {
            dut::Scratch<uint64_t> a((1u << 17) + 1000 * i), b(100 + i);          // one pooled (>= 1 MB), one not
            a.get()[0] = i; a.get()[(1u << 17) + 1000 * i - 1] = i; b.get()[99 + i] = i;
            bytes.fetch_add(a.get()[0] + b.get()[99 + i]);
            dut::Scratch<uint64_t> c(std::move(a));
            if (a || !c) abort();
        });
        // (how many blocks the pool holds now depends on how the threads interleaved: not printed)
        printf("crew: %llu calls over 200 jobs; scratch pool: checksum %llu, %s\n", ran, bytes.load(), dut::scratch_pool().idle.size() <= dut::ScratchPool::kScratchKeep ? "within its limit" : "OVER ITS LIMIT");
    }
    // summary JSON with awkward names
    dut_contig_stats st[2]; memset(st, 0, sizeof(st));
    st[0].length = 20; st[0].n_covered_bases = 12; st[0].summed_coverage = 22; st[0].summed_baseq = 420; st[0].summed_mapq = 960; st[0].quality_bases = 14; st[0].n_reads = 3;
    st[1].length = 0;
    const char *names[2] = {"chr\"T\\\x01", "chrUn_x"};
    uint64_t counts[12] = {2, 2, 8, 6, 0, 2, 0, 0, 0, 0, 0, 0};
    dut_export_meta meta; memset(&meta, 0, sizeof(meta));
    meta.aligner = "BWA"; meta.reference_build = "GRCh38"; meta.sequencing_platform = "NovaSeq"; meta.read_length = 150;
    meta.bed_file = "x.bed"; meta.summary_html = "s.html";
    char *js = nullptr; size_t jl = 0;
    if (dut_coverage_output_json(st, names, counts, 2, &meta, &js, &jl) == CL_OK) printf("json %zu bytes\n", jl);
    dut_free(js);
    // BED writer with the coverage figures + the HTML report (host only), into the tree file's directory
    if (argc > 2) {
        const std::string dir = std::string(argv[2]).substr(0, std::string(argv[2]).find_last_of('/'));
        dut_profiler *pr = dut_profiler_new((dir + "/san.bed").c_str());
        if (pr) {
            dut_profiler_enable_plots(pr, 5000);
            const cl_interval iv1[4] = {{0, 3, CL_REF_N}, {3, 900, CL_CALLABLE}, {900, 2100, CL_POOR_MAPPING_QUALITY}, {2100, 5000, CL_CALLABLE}};
            const uint64_t c1[6] = {3, 3797, 0, 0, 0, 1200};
            dut_profiler_feed_contig(pr, names[0], iv1, 4, c1);
            dut_profiler_finish_plot(pr, names[0], 5000);
            dut_profiler_feed_contig(pr, "chrM", nullptr, 0, c1);            // inherits the previous contig's last line
            size_t nb = 0; uint32_t stride = 0;
            std::vector<uint32_t> a(512), b(512), c(512);
            dut_profiler_plot_bins(pr, "chrM", 16569, &stride, a.data(), b.data(), c.data(), 512, &nb);
            printf("plot: %d figure(s), %zu bins of %u\n", dut_profiler_finish_plot(pr, "chrM", 16569), nb, stride);
            dut_profiler_free(pr);
        }
        printf("html rc %d\n", dut_write_html_report(st, names, counts, 2, &meta, 10000, (dir + "/san.html").c_str()));
    }
    // the orchestration of the several-device call (threads, reader pairs, results handed to the BED writer in tid
    // order) over the stand-in engine above: the BED of 1, 2 and 3 "devices" must be the same text
    if (argc > 3) {
        const std::string dir = std::string(argv[3]).substr(0, std::string(argv[3]).find_last_of('/'));
        unsigned long long first = 0;
        for (int nd = 1; nd <= 3; ++nd) {
            const int devs[3] = {0, 1, 1};
            const std::string bed = dir + "/multi" + std::to_string(nd) + ".bed";
            char e2[512] = {0};
            const int rc = dut_coverage_files_multi(argv[1], argv[3], bed.c_str(), nullptr, nullptr, &opt, nullptr, 0, devs, (size_t)nd, 0u, e2, sizeof(e2));
            std::string text;
            if (FILE *f = fopen(bed.c_str(), "rb")) { char buf[4096]; size_t g; while ((g = fread(buf, 1, sizeof(buf), f)) > 0) text.append(buf, g); fclose(f); }
            const unsigned long long h = checksum(text.data(), text.size());
            if (nd == 1) first = h;
            printf("multi %d device(s): rc %d%s%s, bed %zu bytes %s\n", nd, rc, rc ? " " : "", rc ? e2 : "", text.size(), h == first ? "same" : "DIFFERENT");
            if (rc == CL_OK && h != first) return 4;
        }
        const int bad[2] = {0, 99};
        char e3[512] = {0};
        printf("multi, a device that does not exist: rc %d\n", dut_coverage_files_multi(argv[1], argv[3], (dir + "/multi_bad.bed").c_str(), nullptr, nullptr, &opt, nullptr, 0, bad, 2, 0u, e3, sizeof(e3)));
    }
    return 0;
}
